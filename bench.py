#!/usr/bin/env python3
"""bench.py -- throughput of the SAD stereo matcher on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one rank
per GPU).  A step = one pass of the hot path (sadbm_compute_batch_device: plane prep + matcher +
zero frame) over one batch of synthetic stereo pairs that are already resident in HBM.  Default
workload = BASELINE.json configs[2], the config the metric is quoted on: rds(3,3840,2160,16,96),
15x15 window, D=128, a batch of 16 distinct frames per step (the grid of 512-thread blocks needs a few thousand
blocks to balance; 2 frames per step reach 3.1, 16 reach 4.0 Tdisp-evals/s).  Multi-GPU = batch data parallelism: every rank processes its own batch, no
data-path collective (weak scaling); value = evals of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).  Inputs come from opencv_amd.synth; oracle/ is
touched only by the cpu_baseline leg (rank 0, N=1), which times the CPU port and checks the GPU rows against it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (W, H, win, D, rds seed, d_bg, d_fg, pairs per step)
    # pairs per step: large enough that launch ramp and tail are amortised (scan in DESIGN.md section 7); cfg5 is BASELINE's 512
    "cfg2_1080p_w9_d64": (1920, 1080, 9, 64, 2, 8, 40, 64),
    "cfg2b_1080p_w9_d128": (1920, 1080, 9, 128, 2, 8, 40, 64),
    "cfg3_4k_w15_d128": (3840, 2160, 15, 128, 3, 16, 96, 32),
    "cfg4_8k_w21_d256": (7680, 4320, 21, 256, 4, 32, 200, 4),
    "cfg5_kitti_w7_d128": (1242, 375, 7, 128, 1000, 5, 60, 512),
}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_NOMINAL_TOPS = 39.3         # 256 CU x 64 lanes x 2.4 GHz lane-ops/s (4-cycle wave64 issue)


def cpu_baseline(L, R, win, D, threads, gpu_rows):
    """The ONLY place bench.py touches oracle/: the port of the reference loop (OpenMP over rows) runs on a bounded
    sample -- the top rows of the first frame, sized for ~10-30 s of CPU work -- is timed there, and its output is the
    check of the GPU result on the same rows (`verified_bit_exact`).  The oracle is never the thing measured as `value`."""
    import numpy as np
    import oracle
    H, W = L.shape
    evals_per_row = W * D
    # naive cost ~ win^2 byte-ADs per eval; aim at ~2e10 byte-ADs per thread-second-ish total budget
    budget = 6e9 * threads
    out_rows = int(max(4, min(H - win, budget / (evals_per_row * win * win))))
    rows_in = out_rows + win
    Ls, Rs = L[:rows_in].copy(), R[:rows_in].copy()
    t0 = time.perf_counter()
    exp = oracle.literal_mt(Ls, Rs, win, D, 0, threads)
    dt = time.perf_counter() - t0
    verified = bool(np.array_equal(gpu_rows(out_rows), exp[:out_rows]))
    evals = (rows_in) * W * D      # metric counts full W*H*D of the frame handed in
    return verified, {"value": round(evals / dt / 1e9, 4), "unit": "Gdisp-evals/s", "cores": threads, "kind": "port",
                      "sample": "oracle.literal_mt on the top %d rows (%d output rows) of the same %dx%d frame, win=%d D=%d, %.1f s; "
                                "its output is also the bit-exact check of the GPU rows" % (rows_in, out_rows, W, H, win, D, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3_4k_w15_d128", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per step per GPU (0 = workload default)")
    ap.add_argument("--flags", type=int, default=0, help="SADBM_* flag bits (4 = L1-operand matcher variant)")
    ap.add_argument("--split", default="batch", choices=["batch", "bands"],
                    help="batch: every rank its own pairs, no collective (weak scaling); bands: ONE frame split into "
                         "row bands + one RCCL all-gather per frame (strong scaling, BASELINE config 4)")
    ap.add_argument("--no-verify", action="store_true", help="kept for old command lines; verification lives in the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (and with it the bit-exact check)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import opencv_amd as oa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # SADBM_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices);
    # the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("SADBM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    W, H, win, D, seed, dbg, dfg, pairs = WORKLOADS[args.workload]
    if args.pairs > 0:
        pairs = args.pairs

    from opencv_amd import synth       # product-side generator of the SURVEY 8(d) stereograms
    Ls, Rs = [], []
    for i in range(pairs):
        # distinct pairs per slot and per rank (same generator family as BASELINE.md)
        Li, Ri = synth.rds(seed + 7919 * i + 104729 * rank, W, H, dbg, dfg, 0)
        Ls.append(Li)
        Rs.append(Ri)
    ctx = oa.Context(local)
    bands = args.split == "bands"
    if bands:
        # one frame (seed without the rank term so all ranks split the SAME frame); rank keeps band + halo only
        pairs = 1
        Lf, Rf = synth.rds(seed, W, H, dbg, dfg, 0)
        Ls, Rs = [Lf], [Rf]
        (r0, r1, per), lb, rb = oa.scatter_band(torch.from_numpy(Lf), torch.from_numpy(Rf), rank, world, win)
        dLb, dRb = lb.contiguous().to(dev), rb.contiguous().to(dev)
        full = [None]

        def compute_band(lbt, rbt, frame_rows, row0, rows_out, w, d, flags, out):
            return oa.compute_band_device(lbt, rbt, frame_rows, row0, rows_out, w, d, flags, out=out, ctx=ctx)

        def step():
            full[0] = oa.disparity_row_bands(dLb, dRb, H, win, D, rank, world, flags=args.flags, compute_band=compute_band)
    else:
        dL = torch.from_numpy(np.stack(Ls)).to(dev)
        dR = torch.from_numpy(np.stack(Rs)).to(dev)
        dO = torch.empty((pairs, H, W), dtype=torch.uint8, device=dev)

        def step():
            oa.compute_batch_device(dL, dR, win, D, args.flags, out=dO, ctx=ctx)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nl, matcher_ms, prep_ms = ctx.profile_end()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    evals_step = W * H * D * pairs                     # per GPU per step (metric counts full W*H*D)
    if bands:
        value = evals_step * args.steps / dt / 1e9     # one frame per step for the whole job
    else:
        value = evals_step * world * args.steps / dt / 1e9
    k_ms = matcher_ms / max(nl, 1)                     # matcher kernel, average launch duration (HIP events)
    frac_rows = (r1 - r0) / H if bands else 1.0        # a band launch processes its share of the frame
    alg_bytes = 3.0 * W * H * pairs * frac_rows        # 2 plane reads + 1 plane write per frame
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    valu_ops = 7.0 * W * H * D * pairs * frac_rows
    traffic = None
    try:
        with open(os.path.join(ROOT, "bench_traffic.json")) as f:
            tj = json.load(f)
        ent = tj.get("%s|%s|pairs=%d" % (args.workload, ctx.last_kernel(), pairs))
        if ent and not bands:
            traffic = ent["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    line = {
        "metric": "Gdisp-evals/sec (WxHxD/s)", "value": round(value, 3), "unit": "Gdisp-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if bands else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: rds(%d,%d,%d,%d,%d) %dx%d window D=%d, %d pairs/step/GPU, device-resident planes"
                               % (args.workload, seed, W, H, dbg, dfg, win, win, D, pairs),
                   "parallelism": ("row-bands%d+allgather" if bands else "batch-dp%d") % world, "kernel": ctx.last_kernel()},
        "frames_per_s": round((1 if bands else pairs * world) * args.steps / dt, 2),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes": alg_bytes,
                     "kernel": ctx.last_kernel(), "kernel_ms": round(k_ms, 4), "prep_ms": round(prep_ms / max(nl, 1), 4),
                     "note": "path is VALU-bound (SURVEY 8d): algorithmic bytes 3*W*H per frame; see valu"},
        "valu": {"ops_per_eval": 7, "achieved_Tops": round(valu_ops / (k_ms * 1e-3) / 1e12, 3) if k_ms > 0 else 0,
                 "nominal_peak_Tops": VALU_NOMINAL_TOPS,
                 "frac": round(valu_ops / (k_ms * 1e-3) / 1e12 / VALU_NOMINAL_TOPS, 4) if k_ms > 0 else 0},
        "verified_bit_exact": None,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu:
            def gpu_rows(n):
                return (full[0] if bands else dO[0])[:n].cpu().numpy()
            verified, line["cpu_baseline"] = cpu_baseline(Ls[0], Rs[0], win, D, min(os.cpu_count() or 1, 16), gpu_rows)
            line["verified_bit_exact"] = verified
            if not verified:
                raise SystemExit("bench: GPU output differs from the oracle -- refusing to report a wrong kernel")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
