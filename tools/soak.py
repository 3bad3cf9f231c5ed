#!/usr/bin/env python3
"""One-off parity soak on the GPU box: random large-ish batches (several column blocks, bands and plane groups) against the
oracle.  tools/soak.py [cases] [seed] [wide]   (wide: windows of 26..255 columns)   -- prints one line per case, exits non-zero on the first mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import opencv_amd as oa
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
t0 = time.time()
for case in range(n):
    win = int(rng.integers(4, 26)) if rng.random() < 0.85 else int(rng.choice([1, 2, 3, 26, 29, 31]))
    D = int(rng.choice([8, 16, 24, 30, 32, 33, 48, 64, 65, 100, 128, 129, 192, 255, 256, 257, 300, 520]))   # beyond 256: the reference's encoding only
    W = int(rng.choice([int(rng.integers(300, 1300)), int(rng.integers(1300, 4200)), int(rng.integers(40, 300)), 1920, 1242, 2048, 640, 384, 450]))
    if wide:
        win = int(rng.integers(26, 256))
    H = int(rng.integers(win + 2, win + 200 if wide else 420))
    pairs = int(rng.choice([1, 2, 3, 5, 9]))
    flags = int(rng.choice([0, 0, oa.KERNEL_STRIP, oa.KERNEL_QUAD])) | (int(rng.choice([0, oa.RAW_LOC])) if D <= 256 else 0)
    ch = int(rng.choice([1, 1, 1, 3, 4]))                      # interleaved colour frames: the cost is channel 0's (SAD.cpp:19-20)
    Ls, Rs = [], []
    for i in range(pairs):
        g = np.random.default_rng(1000 * case + i)
        L = g.integers(0, 256, (H, W), dtype=np.uint8)
        R = g.integers(0, 256, (H, W), dtype=np.uint8)
        sh = int(g.integers(1, max(2, min(D, W // 2))))
        R[:, :W - sh] = L[:, sh:]
        if case % 5 == 4:
            L = np.where(L & 0x80, 255, 0).astype(np.uint8); R = np.where(R & 0x80, 255, 0).astype(np.uint8)
        if ch > 1:
            L = np.ascontiguousarray(np.stack([L] + [g.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(ch - 1)], -1))
            R = np.ascontiguousarray(np.stack([R] + [g.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(ch - 1)], -1))
        Ls.append(L); Rs.append(R)
    ctx = oa.Context(0)
    out = oa.compute_batch_device(torch.from_numpy(np.stack(Ls)).cuda(), torch.from_numpy(np.stack(Rs)).cuda(), win, D, flags, ctx=ctx).cpu().numpy()
    of = oracle.RAW_LOC if flags & oa.RAW_LOC else 0
    for i in range(pairs):
        exp = oracle.box(np.ascontiguousarray(Ls[i][..., 0]) if ch > 1 else Ls[i], np.ascontiguousarray(Rs[i][..., 0]) if ch > 1 else Rs[i], win, D, of)
        if not np.array_equal(out[i], exp):
            bad = np.argwhere(out[i] != exp)
            print("MISMATCH", case, (win, D, W, H, ch, pairs, flags), i, len(bad), bad[:5], flush=True)
            sys.exit(1)
    print("ok %3d win=%3d D=%3d %4dx%3d ch=%d pairs=%d flags=%2d %s  t=%.0fs" % (case, win, D, W, H, ch, pairs, flags, ctx.last_kernel(), time.time() - t0), flush=True)
print("soak passed:", n, "cases")
