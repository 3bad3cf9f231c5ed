#!/usr/bin/env python3
"""Rehearsal of disparity_row_bands with sub-bands under gloo, ranks sharing one GPU (GPU box):
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/gloo_bands_check.py"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(110, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
import opencv_amd as oa
from opencv_amd import synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
W, H, win, D = 1920, 1080, 9, 64
L, R = synth.rds(5, W, H, 8, 40, 0)
(r0, r1, per), lb, rb = oa.scatter_band(torch.from_numpy(L), torch.from_numpy(R), rank, world, win)
lb, rb = lb.contiguous().cuda(), rb.contiguous().cuda()
ref = None
for sb in (1, 2, 4, 7):
    for dst in (None, 0):
        print("rank", rank, "sub_bands", sb, "dst", dst, "start", flush=True)
        full = oa.disparity_row_bands(lb, rb, H, win, D, rank, world, sub_bands=sb, dst=dst)
        torch.cuda.synchronize()
        if full is not None:
            h = int(full.to(torch.int64).sum().item())
            ref = ref or h
            assert h == ref, (sb, dst)
        print("rank", rank, "sub_bands", sb, "dst", dst, "ok", flush=True)
dist.barrier()
print("rank", rank, "done", flush=True)
