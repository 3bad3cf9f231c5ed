"""CPU tests of the multi-GPU host logic: band arithmetic and the world_size-2 all-gather path on the
gloo backend.  The per-rank compute function is the CPU oracle here (checker standing in for the GPU
so that the partition + collective code runs without a GPU); the GPU tests run the same code with
the HIP path on one device."""
import os
import socket

import numpy as np
import pytest

import opencv_amd as oa
import oracle
from helpers import noise_pair


@pytest.mark.parametrize("rows", [1, 7, 64, 83, 4320])
@pytest.mark.parametrize("nb", [1, 2, 3, 4, 8])
def test_band_ranges_cover_exactly(rows, nb):
    per, ranges = oa.band_ranges(rows, nb)
    assert len(ranges) == nb and per * nb >= rows
    covered = []
    for (r0, r1) in ranges:
        assert 0 <= r0 <= r1 <= rows and r1 - r0 <= per
        covered += list(range(r0, r1))
    assert covered == list(range(rows))


@pytest.mark.parametrize("rows", [0, 1, 7, 83, 4320])
@pytest.mark.parametrize("nb", [1, 2, 3, 8])
def test_c_abi_band_plan_equals_the_python_partition(rows, nb):
    """sadbm_band_plan (what sadbm_compute_multi splits a frame with) == band_ranges / band_input_rows."""
    per, ranges = oa.band_ranges(rows, nb)
    for i, (r0, r1) in enumerate(ranges):
        for win in (1, 7, 21):
            a, b, c = oa.band_plan(rows, nb, win, i)
            assert (a, b) == (r0, r1)
            assert c == (oa.band_input_rows(r0, r1, rows, win)[1] if r1 > r0 else r0)


def test_band_halo_is_downward_only():
    assert oa.band_input_rows(540, 1080, 4320, 21) == (540, 1100)
    assert oa.band_input_rows(3780, 4320, 4320, 21) == (3780, 4320)


@pytest.mark.parametrize("nb", range(1, 9))
def test_bands_of_oracle_concatenate(nb):
    """per-band restatement with the frame-row loop bound == single call, H not divisible."""
    H, W, win, D = 45, 60, 7, 20
    L, R = noise_pair(nb, H, W, shift=3)
    full = oracle.literal(L, R, win, D)
    per, ranges = oa.band_ranges(H, nb)
    parts = []
    for (r0, r1) in ranges:
        i0, i1 = oa.band_input_rows(r0, r1, H, win)
        # a band only sees its own rows + halo; the loop bound of SAD.cpp:34 uses the frame row index
        parts.append(oracle.band(L[i0:i1], R[i0:i1], H, r0, r1 - r0, win, D))
    assert np.array_equal(np.concatenate(parts, 0), full)


def test_shard_pairs_partition():
    for world in (1, 2, 4, 8):
        got = sorted(sum((oa.shard_pairs(512, r, world) for r in range(world)), []))
        assert got == list(range(512))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, win, D, q, sub_bands=1, dst=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, R = noise_pair(99, H, W, shift=4)

    def cpu_band(lb, rb, frame_rows, row0, rows_out, w, d, flags, out):
        # checker standing in for sadbm_compute_band_device on a CPU tensor
        out.copy_(torch.from_numpy(oracle.band(lb.numpy(), rb.numpy(), frame_rows, row0, rows_out, w, d)))
        return out

    (r0, r1, per), lb, rb = oa.scatter_band(torch.from_numpy(L), torch.from_numpy(R), rank, world, win)
    full = oa.disparity_row_bands(lb.contiguous(), rb.contiguous(), H, win, D, rank, world, compute_band=cpu_band, sub_bands=sub_bands, dst=dst)
    q.put((rank, None if full is None else full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [40, 41])
def test_row_bands_all_gather_gloo_world2(H):
    import torch.multiprocessing as mp
    W, win, D, world = 52, 7, 16, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, win, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    L, R = noise_pair(99, H, W, shift=4)
    exp = oracle.literal(L, R, win, D)
    for r in range(world):
        assert np.array_equal(outs[r], exp), r


@pytest.mark.parametrize("H,sub_bands,dst,world", [(41, 3, None, 2), (40, 20, None, 2), (37, 2, 1, 2), (9, 4, 0, 2),
                                                   (45, 4, None, 2),      # per = 23: pieces of 6, 6, 6, 5 rows, rank 1's band one row short
                                                   (47, 5, None, 3),      # three ranks, per = 16: pieces 4, 4, 4, 4 (+ none), last band 15 rows
                                                   (50, 7, 2, 3),         # seven outstanding pieces gathered to rank 2 only, last piece 2 rows
                                                   (23, 6, None, 4)])     # four ranks, per = 6: six one-row pieces, the last band 5 rows
def test_row_bands_pipelined_gather_and_gather_to_one_gloo_world2(H, sub_bands, dst, world):
    """sub_bands > 1: pieces of the band are gathered asynchronously while the next one is computed (on CPU tensors every piece
    stays outstanding until the end: 4-20 async works in flight); dst: only that rank receives the map.  Bands shorter than
    the window, sub-bands of a single row, uneven last pieces and last bands, 2-4 ranks."""
    import torch.multiprocessing as mp
    W, win, D = 52, 7, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, win, D, q, sub_bands, dst)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    L, R = noise_pair(99, H, W, shift=4)
    exp = oracle.literal(L, R, win, D)
    for r in range(world):
        if dst is None or r == dst:
            assert np.array_equal(outs[r], exp), r
        else:
            assert outs[r] is None
