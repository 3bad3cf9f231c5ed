#!/usr/bin/env python3
"""Whole-step time (call + synchronise) of the default routing against SADBM_DIRECT=1 (frames read in place wherever possible): where
the plane-reading variants with their prep pass still pay.  tools/step_direct_ab.py ["W,H,pairs,win,D" ...]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import opencv_amd as oa
from opencv_amd import synth
jobs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1920,1080,64,9,64),(1920,1080,64,13,64),(1920,1080,64,11,48),(1920,1080,64,5,64),(3840,2160,16,9,64),(1920,1080,8,9,64)]
for (W,H,pairs,win,D) in jobs:
    L,R = synth.rds(7,W,H,4,20,0)
    dL = torch.from_numpy(np.stack([L]*pairs)).cuda(); dR = torch.from_numpy(np.stack([R]*pairs)).cuda()
    dO = torch.empty((pairs,H,W),dtype=torch.uint8,device="cuda")
    line = "%dx%d x%d w%d D%d:" % (W,H,pairs,win,D)
    for mode in ("", "1", "", "1"):
        if mode: os.environ["SADBM_DIRECT"] = mode
        else: os.environ.pop("SADBM_DIRECT", None)
        oa.reload_env()
        ctx = oa.Context(0)
        for _ in range(5): oa.compute_batch_device(dL,dR,win,D,0,out=dO,ctx=ctx)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10): oa.compute_batch_device(dL,dR,win,D,0,out=dO,ctx=ctx)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter()-t0)/10*1e3)
        line += "  %s %.4f (%s)" % ("direct" if mode else "default", best, ctx.last_kernel())
    print(line, flush=True)
