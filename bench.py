#!/usr/bin/env python3
"""bench.py -- throughput of the SAD stereo matcher on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  N > 1 runs one rank per GPU: either launched by
`python -m torch.distributed.run ... bench.py --gpus N ...` (RANK / WORLD_SIZE in the environment) or, when started
plainly, bench.py itself starts that launcher as a CHILD process before anything touches a GPU.  WORLD_SIZE must equal
--gpus.

A step = one pass of the hot path (sadbm_compute_batch_device: plane prep + matcher + zero frame) over one batch of
synthetic stereo pairs already resident in HBM.  Default workload = BASELINE.json configs[2], the config the metric is
quoted on: rds(3,3840,2160,16,96), 15x15 window, D=128, 32 distinct frames per step.  Multi-GPU = batch data
parallelism: every rank processes its own batch, no data-path collective (weak scaling); value = disparity
evaluations of all ranks / max-over-ranks time.  `--split bands` is BASELINE config 4's form instead: ONE frame split
into row bands, one RCCL all-gather per frame (strong scaling).

Prints ONE JSON line on rank 0:
  value / ms_per_step   whole-job throughput of the timed steps
  roofline              the BINDING roof: VALU issue.  achieved = 7*W*H*D lane-ops per frame x frames per
                        launch / the matcher kernel's average duration (HIP events on the launch stream); peak = the
                        measured issue rate of the instruction class the kernel is made of; the full-rate class and the
                        HBM roof (3*W*H bytes per frame, PMC traffic) are stated beside it
  latency / frames_* / cfg4_* / cfg5_*   one-frame call latency and one leg per other BASELINE configuration (N = 1 only), each with
                        its kernel, kernel_ms, VALU fraction and a comparison of its output with the oracle
  cpu_baseline / cpu_baseline_1core   the CPU port of the reference loop on this box's host cores (N = 1 only)
  verified_bit_exact    the last step compared with the CPU oracle, outside the timed region: every pair at N = 1; at N > 1 rank 0
                        checks its own pairs (batch) or the gathered map against oracle.box of the whole frame (bands)
  ranks                 N > 1: per-rank time of the timed region (min / max), and for --split bands the band compute and the
                        gather timed separately after the timed region
Inputs come from opencv_amd.synth; oracle/ is touched only after the timed region (verification + CPU baselines).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (W, H, win, D, rds seed, d_bg, d_fg, pairs per step)
    # pairs per step: large enough that launch ramp and tail are amortised (scan in DESIGN.md section 7); cfg5 is BASELINE's 512
    "cfg1_tsukuba_size_w7_d30": (384, 288, 7, 30, 1, 3, 14, 256),
    "cfg2_1080p_w9_d64": (1920, 1080, 9, 64, 2, 8, 40, 64),
    "cfg2b_1080p_w9_d128": (1920, 1080, 9, 128, 2, 8, 40, 64),
    "cfg2c_1080p_w7_d30": (1920, 1080, 7, 30, 2, 4, 20, 64),
    "cfg3_4k_w15_d128": (3840, 2160, 15, 128, 3, 16, 96, 32),
    "cfg3c_4k_w7_d32": (3840, 2160, 7, 32, 3, 4, 24, 32),
    "cfg4_8k_w21_d256": (7680, 4320, 21, 256, 4, 32, 200, 4),
    "cfg5_kitti_w7_d128": (1242, 375, 7, 128, 1000, 5, 60, 512),
    "wide_1080p_w51_d64": (1920, 1080, 51, 64, 2, 8, 40, 16),      # beyond the strip / quad matchers: sad_wide_kernel (profiling only)
}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU issue, lane-ops/s = 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / (cycles per wave64 instruction and SIMD):
#   full-rate class, 2 cycles (MI355X_MICROARCH.md constants table; measured v_add_u32 2.3)      -> 78.6 T
#   the class the matcher is made of -- v_sad_u8 / v_sad_hi_u8 / v_perm_b32 / v_min3_u32 / v_pk_* / v_mad_u32_u16 /
#   v_add3_u32, measured 4.2-4.4 cycles (profiles/r01_ubench_instruction_rates.txt)              -> 39.3 T
VALU_PEAK_SAD_CLASS = 39.3
VALU_PEAK_FULL_RATE = 78.6
OPS_PER_EVAL = 7                 # algorithmic lane-ops per disparity evaluation


def sha(a):
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a child (fresh processes; this
    process has not touched a GPU and never will) and hand its exit status on."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def cpu_legs(L, R, win, D, threads, gpu_frame):
    """The CPU port of the reference loop (oracle.literal_mt = SAD.cpp:34-59 with OpenMP over rows; oracle.literal = the
    same on ONE thread, as the reference runs) on bounded samples -- the top rows of the first frame of the batch --
    each also compared with the GPU rows."""
    import numpy as np
    import oracle
    H, W = L.shape
    legs = {}
    ok = True
    for key, thr, budget in (("cpu_baseline", threads, 6e9 * threads), ("cpu_baseline_1core", 1, 5e9)):
        out_rows = int(max(4, min(H - win, budget / (W * D * win * win))))
        rows_in = out_rows + win
        Ls, Rs = L[:rows_in].copy(), R[:rows_in].copy()
        t0 = time.perf_counter()
        exp = oracle.literal_mt(Ls, Rs, win, D, 0, thr) if thr > 1 else oracle.literal(Ls, Rs, win, D, 0)
        dt = time.perf_counter() - t0
        ok = ok and bool(np.array_equal(gpu_frame[:out_rows], exp[:out_rows]))
        legs[key] = {"value": round(rows_in * W * D / dt / 1e9, 4), "unit": "Gdisp-evals/s", "cores": thr, "kind": "port",
                     "host_cpus": os.cpu_count(),
                     "sample": "oracle.%s (port of the reference loop, O(win^2) per evaluation) on the top %d rows (%d output rows) of the same "
                               "%dx%d frame, win=%d D=%d, %.1f s" % ("literal_mt" if thr > 1 else "literal", rows_in, out_rows, W, H, win, D, dt)}
    return ok, legs


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
    ap.add_argument("--sub-bands", type=int, default=1, help="--split bands: compute the band in this many pieces and gather each "
                                                              "finished piece asynchronously while the next one is computed")
    ap.add_argument("--no-verify", action="store_true", help="skip the whole-batch comparison with the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines, the verification and the extra single-GPU legs")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency and 1080p legs")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))                    # before any GPU call; children are fresh processes
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: WORLD_SIZE=%d but --gpus %d -- refusing to mislabel the run" % (world, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import opencv_amd as oa

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
            full[0] = oa.disparity_row_bands(dLb, dRb, H, win, D, rank, world, flags=args.flags, compute_band=compute_band,
                                             sub_bands=args.sub_bands)
    else:
        Ls, Rs = [], []
        for i in range(pairs):
            # distinct pairs per slot and per rank (same generator family as BASELINE.md)
            Li, Ri = synth.rds(seed + 7919 * i + 104729 * rank, W, H, dbg, dfg, 0)
            Ls.append(Li)
            Rs.append(Ri)
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
    ranks_info = None
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        mine = torch.tensor([dt, matcher_ms / max(nl, 1)], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dts = [float(a[0].item()) for a in allr]
        kms = [float(a[1].item()) for a in allr]
        dt = max(dts)                                      # the contract's time: the slowest rank
        ranks_info = {"ms_per_step_min": round(min(dts) / args.steps * 1e3, 4), "ms_per_step_max": round(max(dts) / args.steps * 1e3, 4),
                      "kernel_ms_min": round(min(kms), 4), "kernel_ms_max": round(max(kms), 4)}
        if bands:
            # compute and exchange separately (after the timed region): the band into this rank's slot, then the gather alone
            per_rows = -(-H // world)
            sendb = torch.zeros((per_rows, W), dtype=torch.uint8, device=dev)
            fullb = torch.empty((world * per_rows, W), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(); dist.barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                if r1 > r0:
                    compute_band(dLb, dRb, H, r0, r1 - r0, win, D, args.flags, sendb[: r1 - r0])
            torch.cuda.synchronize()
            tc = (time.perf_counter() - t1) / args.steps
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                if backend == "nccl":
                    dist.all_gather_into_tensor(fullb, sendb)
                else:
                    hs = sendb.cpu(); ho = [torch.empty_like(hs) for _ in range(world)]
                    dist.all_gather(ho, hs)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - t1) / args.steps
            tt = torch.tensor([tc, tg], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ranks_info["band_compute_ms_max"] = round(float(tt[0].item()) * 1e3, 4)
            ranks_info["gather_ms_max"] = round(float(tt[1].item()) * 1e3, 4)
            ranks_info["gather_note"] = "all_gather_into_tensor of %d x %d u8 rows per rank, timed alone" % (per_rows, W) if backend == "nccl" \
                else "rehearsal backend %s: pieces through host memory" % backend

    evals_step = W * H * D * pairs                     # per GPU per step (metric counts full W*H*D)
    if bands:
        value = evals_step * args.steps / dt / 1e9     # one frame per step for the whole job
    else:
        value = evals_step * world * args.steps / dt / 1e9
    k_ms = matcher_ms / max(nl, 1)                     # matcher kernel, average launch duration (HIP events)
    frac_rows = (r1 - r0) / H if bands else 1.0        # a band launch processes its share of the frame
    alg_bytes = 3.0 * W * H * pairs * frac_rows        # 2 plane reads + 1 plane write per frame
    hbm_achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    valu_ops = float(OPS_PER_EVAL) * W * H * D * pairs * frac_rows
    valu_achieved = valu_ops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    # HBM traffic of the kernel that just ran: PMC passes cannot run inside this process, so the figure comes from the committed
    # rocprofv3 summaries, keyed by (workload, kernel, pairs per launch).  A default-size run whose key is missing means the
    # kernel changed without a new PMC pass: say so instead of quoting nothing (or somebody else's bytes).
    traffic, traffic_src = None, None
    with open(os.path.join(ROOT, "bench_traffic.json")) as f:
        tj = json.load(f)
    tkey = "%s|%s|pairs=%d" % (args.workload, ctx.last_kernel(), pairs)
    ent = tj.get(tkey)
    if not bands:
        if ent:
            traffic, traffic_src = ent["hbm_bytes_per_launch"], ent["source"]
        elif args.pairs == 0 and args.flags == 0 and not os.environ.get("SADBM_BENCH_ALLOW_NO_TRAFFIC"):
            raise SystemExit("bench: bench_traffic.json has no entry '%s' -- the kernel or batch size changed; run tools/prof.sh on this "
                             "workload and add FETCH_SIZE / WRITE_SIZE of the new kernel (or set SADBM_BENCH_ALLOW_NO_TRAFFIC=1)" % tkey)
    line = {
        "metric": "Gdisp-evals/sec (WxHxD/s)", "value": round(value, 3), "unit": "Gdisp-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if bands else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: rds(%d,%d,%d,%d,%d) %dx%d window D=%d, %d pairs/step/GPU, device-resident planes"
                               % (args.workload, seed, W, H, dbg, dfg, win, win, D, pairs),
                   "parallelism": ("row-bands%d+allgather" if bands else "batch-dp%d") % world, "kernel": ctx.last_kernel()},
        "frames_per_s": round((1 if bands else pairs * world) * args.steps / dt, 2),
        # the binding roof: VALU issue
        "roofline": {"bound": "valu", "achieved": round(valu_achieved, 3), "peak": VALU_PEAK_SAD_CLASS, "unit": "Tlane-ops/s",
                     "frac": round(valu_achieved / VALU_PEAK_SAD_CLASS, 4),
                     "traffic": traffic,
                     "algorithmic_ops": valu_ops, "ops_per_eval": OPS_PER_EVAL,
                     "kernel": ctx.last_kernel(), "kernel_ms": round(k_ms, 4), "prep_ms": round(prep_ms / max(nl, 1), 4),
                     "peak_note": "39.3 T = measured issue rate of the 4-cycle VALU class the matcher consists of (v_sad_u8, v_perm_b32, "
                                  "v_min3_u32, v_pk_*: 4.2 cycles per wave64 and SIMD); 2-cycle class (v_add_u32 ...) 78.6 T",
                     "peak_full_rate": VALU_PEAK_FULL_RATE, "frac_full_rate": round(valu_achieved / VALU_PEAK_FULL_RATE, 4),
                     "traffic_note": "traffic = HBM bytes per launch from the PMC passes (bench_traffic.json); algorithmic bytes below",
                     "traffic_source": traffic_src},
        # the secondary roof north_star asks for: HBM, algorithmic bytes 3*W*H per frame -- ~1 % of peak, by construction
        "hbm": {"bound": "hbm", "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes": alg_bytes, "traffic": traffic},
        "verified_bit_exact": None,
    }
    if ranks_info:
        line["ranks"] = ranks_info
    if rank == 0 and world > 1 and not args.no_verify:
        # N > 1 lines verify themselves: rank 0 checks its own last step (batch: its pairs, at most 8 of them so that the other
        # ranks do not wait long; bands: the gathered map against the single-call result of the whole frame, SAD.cpp:34-59)
        import oracle
        t0 = time.perf_counter()
        if bands:
            okv = sha(full[0].cpu().numpy()) == sha(oracle.box(Ls[0], Rs[0], win, D))
            nchk = 1
        else:
            nchk = min(pairs, 8)
            idx = [pairs - 1 - i for i in range(nchk)]
            okv = all(sha(dO[i].cpu().numpy()) == sha(oracle.box(Ls[i], Rs[i], win, D)) for i in idx)
        line["verified_bit_exact"] = bool(okv)
        line["verified"] = ("rank 0: the gathered map equals oracle.box of the whole frame (sha256), %.1f s" if bands else
                            "rank 0: the last %d of its %d frames of the last step equal oracle.box (sha256), %%.1f s" % (nchk, pairs)) % (time.perf_counter() - t0)
        if not okv:
            raise SystemExit("bench: rank 0's output differs from the oracle -- refusing to report a wrong kernel")
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle
        # (1) every pair of the last step against the sliding-sum oracle, whole frames, outside the timed region
        if not args.no_verify:
            t0 = time.perf_counter()
            bad = []
            if bands:
                if sha(full[0].cpu().numpy()) != sha(oracle.box(Ls[0], Rs[0], win, D)):
                    bad.append(0)
            else:
                for i in range(pairs):
                    if sha(dO[i].cpu().numpy()) != sha(oracle.box(Ls[i], Rs[i], win, D)):
                        bad.append(i)
            line["verified_bit_exact"] = not bad
            line["verified"] = "%d of %d whole frames of the last step equal oracle.box (sha256), %.1f s" % (
                pairs - len(bad), pairs, time.perf_counter() - t0)
            if bad:
                raise SystemExit("bench: GPU output differs from the oracle on pairs %s -- refusing to report a wrong kernel" % bad[:8])
        # (2) CPU baselines beside it (bounded samples), each also checked against the GPU rows
        gpu0 = (full[0] if bands else dO[0]).cpu().numpy()
        ok, legs = cpu_legs(Ls[0], Rs[0], win, D, min(os.cpu_count() or 1, 16), gpu0)
        line.update(legs)
        if not ok:
            raise SystemExit("bench: GPU rows differ from the CPU port -- refusing to report a wrong kernel")
        # (3) what else the metric names: one-frame latency of this workload, and 1080p frames/s at D = 128
        if not args.no_extras and not bands:
            one_l, one_r = dL[:1].contiguous(), dR[:1].contiguous()
            one_o = torch.empty((1, H, W), dtype=torch.uint8, device=dev)
            lat = []
            for i in range(25):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                oa.compute_batch_device(one_l, one_r, win, D, args.flags, out=one_o, ctx=ctx)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            ctx.profile_begin(8)
            for i in range(8):
                oa.compute_batch_device(one_l, one_r, win, D, args.flags, out=one_o, ctx=ctx)
            torch.cuda.synchronize()
            n1, m1, p1 = ctx.profile_end()
            lat_ms = sorted(lat[5:])[len(lat[5:]) // 2] * 1e3
            line["latency"] = {"frames": 1, "call_to_sync_ms": round(lat_ms, 4), "matcher_kernel_ms": round(m1 / max(n1, 1), 4),
                               "prep_kernel_ms": round(p1 / max(n1, 1), 4),
                               "value": round(W * H * D / (lat_ms * 1e-3) / 1e9, 1), "unit": "Gdisp-evals/s"}
            # the other BASELINE configurations and the reference's own parameters (SAD.cpp:9,11: winSize 7, DSR 30), 10 steps each:
            # kernel, kernel time (HIP events), its VALU fraction, and frames of the last step compared with the oracle
            for key, wl, n2, nver in (("frames_1080p_d128", "cfg2b_1080p_w9_d128", 64, 1), ("frames_1080p_d64", "cfg2_1080p_w9_d64", 64, 1),
                                      ("frames_1080p_w7_d30", "cfg2c_1080p_w7_d30", 64, 1), ("frames_384x288_w7_d30", "cfg1_tsukuba_size_w7_d30", 256, 4),
                                      ("frames_384x288_w7_d30_x2048", "cfg1_tsukuba_size_w7_d30", 2048, 4),   # the same frames, a launch of 1.6 ms instead of 0.25
                                      ("cfg4_8k_w21_d256", "cfg4_8k_w21_d256", 4, 1), ("cfg5_kitti_w7_d128", "cfg5_kitti_w7_d128", 512, 16)):
                if wl == args.workload:
                    continue
                w2, h2, win2, d2, seed2, bg2, fg2, _n = WORKLOADS[wl]
                fr = [synth.rds(seed2 + 7919 * i, w2, h2, bg2, fg2, 0) for i in range(n2)]
                l2 = torch.from_numpy(np.stack([f[0] for f in fr])).to(dev)
                r2 = torch.from_numpy(np.stack([f[1] for f in fr])).to(dev)
                o2 = torch.empty((n2, h2, w2), dtype=torch.uint8, device=dev)
                # untimed warm-up: at least 3 launches and 0.1 s -- the oracle checks between the legs leave the GPU idle for seconds, and
                # three launches of a 1 ms kernel do not bring the clocks back (the leg then reads 8-10 % low)
                t_w = time.perf_counter()
                while True:
                    for _ in range(3):
                        oa.compute_batch_device(l2, r2, win2, d2, 0, out=o2, ctx=ctx)
                    torch.cuda.synchronize()
                    if time.perf_counter() - t_w > 0.1:
                        break
                ctx.profile_begin(10)
                t0 = time.perf_counter()
                for _ in range(10):
                    oa.compute_batch_device(l2, r2, win2, d2, 0, out=o2, ctx=ctx)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
                n3, m3, p3 = ctx.profile_end()
                km = m3 / max(n3, 1)
                chk = [n2 - 1 - (i * (n2 // nver)) for i in range(nver)]
                ok2 = all(sha(o2[i].cpu().numpy()) == sha(oracle.box(fr[i][0], fr[i][1], win2, d2)) for i in chk)
                line[key] = {"frames_per_s": round(n2 * 10 / dt2, 1), "value": round(w2 * h2 * d2 * n2 * 10 / dt2 / 1e9, 1),
                             "unit": "Gdisp-evals/s", "workload": "%dx%d, %dx%d, D=%d, %d frames per step, 10 steps" % (w2, h2, win2, win2, d2, n2),
                             "kernel": ctx.last_kernel(), "kernel_ms": round(km, 4), "prep_ms": round(p3 / max(n3, 1), 4),
                             "valu_frac": round(OPS_PER_EVAL * w2 * h2 * d2 * n2 / (km * 1e-3) / 1e12 / VALU_PEAK_SAD_CLASS, 4) if km > 0 else None,
                             "frames_checked": len(chk), "last_frame_equals_oracle": ok2}
                del l2, r2, o2
                if not ok2:
                    raise SystemExit("bench: %s output differs from the oracle" % wl)
                if wl == "cfg4_8k_w21_d256":
                    # the same frame through the row-band path of the N-GPU split (world 1: one band, no exchange partner)
                    lf, rf = torch.from_numpy(fr[-1][0]).to(dev), torch.from_numpy(fr[-1][1]).to(dev)

                    def cb(lbt, rbt, frame_rows, row0, rows_out, w, d, flags, out):
                        return oa.compute_band_device(lbt, rbt, frame_rows, row0, rows_out, w, d, flags, out=out, ctx=ctx)
                    for _ in range(2):
                        fb = oa.disparity_row_bands(lf, rf, h2, win2, d2, 0, 1, compute_band=cb)
                    torch.cuda.synchronize()
                    ctx.profile_begin(10)
                    t0 = time.perf_counter()
                    for _ in range(10):
                        fb = oa.disparity_row_bands(lf, rf, h2, win2, d2, 0, 1, compute_band=cb)
                    torch.cuda.synchronize()
                    dt3 = time.perf_counter() - t0
                    n4, m4, p4 = ctx.profile_end()
                    km4 = m4 / max(n4, 1)
                    ok3 = sha(fb.cpu().numpy()) == sha(oracle.box(fr[-1][0], fr[-1][1], win2, d2))
                    line["cfg4_bands_world1"] = {"frames_per_s": round(10 / dt3, 1), "value": round(w2 * h2 * d2 * 10 / dt3 / 1e9, 1), "unit": "Gdisp-evals/s",
                                                 "workload": "one %dx%d frame through disparity_row_bands (world 1), %dx%d, D=%d, 10 steps" % (w2, h2, win2, win2, d2),
                                                 "kernel": ctx.last_kernel(), "kernel_ms": round(km4, 4), "prep_ms": round(p4 / max(n4, 1), 4),
                                                 "valu_frac": round(OPS_PER_EVAL * w2 * h2 * d2 / (km4 * 1e-3) / 1e12 / VALU_PEAK_SAD_CLASS, 4) if km4 > 0 else None,
                                                 "last_frame_equals_oracle": ok3}
                    del lf, rf, fb
                    if not ok3:
                        raise SystemExit("bench: cfg4 row-band output differs from the oracle")
                del fr
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
