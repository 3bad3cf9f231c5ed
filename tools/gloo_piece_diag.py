#!/usr/bin/env python3
"""One-shot diagnosis of the round-2 hang (gpurun_out/b2c.log: two gloo ranks sharing GPU 0, bench.py --split bands with
sub-bands on the 7680x4320 frame, device tensors handed to gloo).  Runs the sub-band exchange loop of
opencv_amd.multi.disparity_row_bands on device tensors in ONE of two variants and prints where it is:
   fresh    piece.clone() as input, freshly allocated contiguous outputs, copied into the map afterwards
   aliased  the round-2 code: the piece is a row slice of this rank's slot of the map, the outputs are row slices of the map
A rank stuck for 40 s dumps the Python stack of every thread (faulthandler) and exits.
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/gloo_piece_diag.py fresh|aliased [sync|async]"""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(40, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import opencv_amd as oa
from opencv_amd import synth
from opencv_amd.bands import band_ranges

variant = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "sync"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
W, H, win, D, sub_bands = 7680, 4320, 21, 256, 4
t00 = time.time()
def say(*a): print("[%6.2f] rank %d %s %s:" % (time.time() - t00, rank, variant, mode), *a, flush=True)
L, R = synth.rds(4, W, H, 32, 200, 0)
(r0, r1, per), lb, rb = oa.scatter_band(torch.from_numpy(L), torch.from_numpy(R), rank, world, win)
lb, rb = lb.contiguous().cuda(), rb.contiguous().cuda()
ctx = oa.Context(0)
say("inputs ready")
for it in range(3):                                           # bench.py calls step() back to back without a host sync
    full = torch.zeros((world * per, W), dtype=torch.uint8, device="cuda")
    send = full[rank * per:(rank + 1) * per]
    sper = -(-per // sub_bands)
    works, keep = [], []
    for s in range(sub_bands):
        p0, p1 = s * sper, min((s + 1) * sper, per)
        s0, s1 = r0 + p0, min(r0 + p1, r1)
        if s1 > s0:
            i1 = min(s1 + win - 1, H)
            oa.compute_band_device(lb[s0 - r0:i1 - r0], rb[s0 - r0:i1 - r0], H, s0, s1 - s0, win, D, 0, out=send[p0:p0 + (s1 - s0)], ctx=ctx)
        if variant == "fresh":
            piece = send[p0:p1].clone()
            outs = [torch.empty_like(piece) for _ in range(world)]
        else:
            piece = send[p0:p1]
            outs = [full[r * per + p0:r * per + p1] for r in range(world)]
        say("iter", it, "piece", s, "rows", p1 - p0, "all_gather issue")
        w = dist.all_gather(outs, piece, async_op=True)
        if mode == "sync":
            w.wait(); say("iter", it, "piece", s, "waited")
        else:
            works.append(w)
        if variant == "fresh":
            keep.append((outs, p0, p1))
    for w in works:
        w.wait()
    for outs, p0, p1 in keep:
        for r in range(world):
            full[r * per + p0:r * per + p1].copy_(outs[r])
    say("iter", it, "all pieces waited")
torch.cuda.synchronize()
say("checksum", int(full[:H].to(torch.int64).sum().item()))
dist.barrier()
say("done")
