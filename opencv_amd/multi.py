"""Multi-GPU host logic: one process per GPU, torch.distributed (RCCL).

Two ways the path shards, both without any cost-volume exchange:
  * batch data parallelism  -- pairs are independent; rank r takes pairs r, r+world, ... ; no collective.
  * row bands of one frame  -- rank r computes output rows [r*per, (r+1)*per) from input rows
    [r0, min(r1+win-1, H)) (downward halo only: the window is anchored top-left, SAD.cpp:39), then ONE
    all-gather of equal-sized u8 bands assembles the full disparity map on every rank.

`compute_band` is the per-rank compute function; the default is the HIP path
(opencv_amd.compute_band_device).  Tests on CPU pass a checker-backed callable to exercise the
partition + collective logic with the gloo backend.
"""
from .bands import band_input_rows, band_ranges


def shard_pairs(n_pairs, rank, world):
    """Indices of the stereo pairs rank `rank` processes (round-robin; no data-path collective)."""
    return list(range(rank, n_pairs, world))


def scatter_band(frame_left, frame_right, rank, world, win):
    """Slice (views) of the frame rows rank `rank` needs: its band plus the win-1 row downward halo."""
    rows = frame_left.shape[0]
    per, ranges = band_ranges(rows, world)
    r0, r1 = ranges[rank]
    i0, i1 = band_input_rows(r0, r1, rows, win)
    return (r0, r1, per), frame_left[i0:i1], frame_right[i0:i1]


def disparity_row_bands(left_band, right_band, frame_rows, win, max_disp, rank, world, group=None, flags=0,
                        compute_band=None, sub_bands=1, dst=None):
    """Band compute + one all-gather.  left_band/right_band: this rank's input rows (band + halo) as
    torch uint8 tensors on the rank's device.  Returns the full frame_rows x cols disparity map
    (identical on every rank).

    sub_bands > 1 pipelines the exchange (over xGMI a ring gather of config 4's bands takes about as long as a
    GPU's share of the compute): the band is computed in sub_bands pieces and the all-gather of a finished piece is issued
    asynchronously, so it runs while the next piece is computed.  Every piece is gathered straight into its place in the map.
    dst = a rank: gather to that rank only (the others return None) -- for callers where one GPU or the host consumes the map."""
    import torch
    import torch.distributed as dist
    if compute_band is None:
        from .stereo import compute_band_device as compute_band
    cols = left_band.shape[1]
    per, ranges = band_ranges(frame_rows, world)
    r0, r1 = ranges[rank]
    # equal counts for the all-gather: every rank contributes `per` rows (short/empty bands are zero-padded)
    if world == 1:
        send = torch.zeros((per, cols), dtype=torch.uint8, device=left_band.device)
        if r1 > r0:
            compute_band(left_band, right_band, frame_rows, r0, r1 - r0, win, max_disp, flags, out=send[: r1 - r0])
        return send[:frame_rows]
    if sub_bands <= 1 and dst is None:
        # the plain form: whole band, then ONE all-gather of equal-sized u8 bands
        send = torch.zeros((per, cols), dtype=torch.uint8, device=left_band.device)
        if r1 > r0:
            compute_band(left_band, right_band, frame_rows, r0, r1 - r0, win, max_disp, flags, out=send[: r1 - r0])
        full = torch.empty((world * per, cols), dtype=torch.uint8, device=left_band.device)
        dist.all_gather_into_tensor(full, send, group=group)
        return full[:frame_rows]
    # Ordering under RCCL: compute_band launches on torch's CURRENT stream (stereo.py passes torch.cuda.current_stream()); an
    # async collective first makes RCCL's stream wait for an event recorded on the current stream at the call, so a piece is
    # gathered after the kernels that wrote it; the next compute_band writes OTHER rows of `send`; work.wait() makes the current
    # stream wait for the gather before the piece's rows are copied into the map.  Nothing depends on timing, and the collective
    # sees only freshly allocated contiguous tensors: the piece goes through all_gather_into_tensor into a [world x rows, cols] slab
    # of its own (the one-call form every RCCL user runs), and the slab is copied into the map's row ranges afterwards -- no list
    # outputs, no output aliasing the input (round-2 advisor finding; one extra device copy of the map per frame).
    # Any other backend with device tensors (gloo rehearsals on a one-GPU box) exchanges each piece through host memory and
    # waits for it at once.  Handing gloo the DEVICE tensors hung in round 2 (two ranks sharing GPU 0, 540 x 7680 pieces of
    # the 8K frame); tools/gloo_piece_diag.py reproduced it once with stack dumps (profiles/r03_gloo_device_tensor_hang.txt):
    # both ranks sit inside the third dist.all_gather(...) CALL -- not in wait() -- with each piece waited for before the
    # next is issued and with piece.clone() / freshly allocated outputs, so neither the aliasing nor the async use causes it.
    backend = dist.get_backend(group)
    overlap = backend == "nccl" or not left_band.is_cuda
    via_host = left_band.is_cuda and backend != "nccl"
    root = dst is None or dst == rank
    full = torch.zeros((world * per, cols), dtype=torch.uint8, device=left_band.device) if root else None
    send = torch.zeros((per, cols), dtype=torch.uint8, device=left_band.device)
    sub_bands = max(1, min(int(sub_bands), per))
    sper = -(-per // sub_bands)
    pending = []                                              # (work, slab or list of per-rank tensors, p0, p1)

    def place(got, p0, p1):                                   # piece rows [p0, p1) of every rank's band -> the map
        if root:
            fv = full.view(world, per, cols)
            if isinstance(got, list):
                for r, g in enumerate(got):
                    fv[r, p0:p1].copy_(g)
            else:
                fv[:, p0:p1].copy_(got.view(world, p1 - p0, cols))

    for s in range(sub_bands):
        p0, p1 = s * sper, min((s + 1) * sper, per)           # rows of the band slot, the same on every rank
        if p1 <= p0:
            break
        s0, s1 = r0 + p0, min(r0 + p1, r1)                    # frame rows of this piece that exist on this rank
        if s1 > s0:
            i1 = min(s1 + win - 1, frame_rows)
            compute_band(left_band[s0 - r0:i1 - r0], right_band[s0 - r0:i1 - r0], frame_rows, s0, s1 - s0, win, max_disp, flags,
                         out=send[p0:p0 + (s1 - s0)])
        piece = send[p0:p1]                                   # contiguous rows of `send`
        if via_host:
            hpiece = piece.cpu()
            houts = [torch.empty_like(hpiece) for _ in range(world)] if root else None
            if dst is None:
                dist.all_gather(houts, hpiece, group=group)
            else:
                dist.gather(hpiece, houts, dst=dst, group=group)
            place(houts, p0, p1)
            continue
        if dst is None:
            slab = torch.empty((world * (p1 - p0), cols), dtype=torch.uint8, device=left_band.device)   # rank-major, as the collective concatenates
            work = dist.all_gather_into_tensor(slab, piece, group=group, async_op=True)
            pending.append((work, slab, p0, p1))
        else:
            outs = [torch.empty((p1 - p0, cols), dtype=torch.uint8, device=left_band.device) for _ in range(world)] if root else None
            work = dist.gather(piece, outs, dst=dst, group=group, async_op=True)
            pending.append((work, outs, p0, p1))
        if not overlap:
            w, got, a, b = pending.pop()
            w.wait()
            place(got, a, b)
    for w, got, a, b in pending:
        w.wait()
        place(got, a, b)
    return full[:frame_rows] if root else None
