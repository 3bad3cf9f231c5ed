"""sadbm_multi_* (single-process multi-GPU C ABI) on the one-GPU box: ngpu = 1 through RCCL (ncclCommInitAll of one
rank, in-place ncclAllGather / grouped ncclBroadcast), and 2..8 "devices" that are all GPU 0 with the peer-copy gather
behind the same interface (the partition and the exchange without a cluster)."""
import numpy as np
import pytest

import opencv_amd as oa
import oracle
from helpers import noise_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_multi():
    """ONE single-rank RCCL communicator for the whole module (ncclCommInitAll starts proxy threads and a bootstrap)."""
    m = oa.Multi(1)
    yield m
    m.close()


@pytest.mark.parametrize("sub_bands", [1, 3, 7, 50])
def test_multi_rccl_one_rank(rccl_multi, sub_bands):
    L, R = noise_pair(7, 211, 333, shift=6)
    m = rccl_multi
    for win, D in ((15, 48), (7, 30), (21, 200)):
        exp = oracle.box(L, R, win, D)
        got = m.compute(L, R, win, D, sub_bands=sub_bands)
        assert np.array_equal(got, exp), (win, D)
        assert np.array_equal(m.device_map(0, *L.shape).cpu().numpy(), exp)


@pytest.mark.parametrize("n", [2, 3, 5, 8])
@pytest.mark.parametrize("sub_bands", [1, 4, 7])
def test_multi_copy_gather_virtual_devices(n, sub_bands):
    """H not divisible by n, bands shorter than the window, empty last bands, BGR input."""
    m = oa.Multi(n, devices=[0] * n, multi_flags=oa.MULTI_COPY_GATHER)
    for (H, W, ch) in ((97, 150, 1), (11, 90, 1), (64, 257, 3)):
        L, R = noise_pair(H + n, H, W, shift=4, ch=ch)
        for win, D in ((9, 40), (15, 128)):
            exp = oracle.box(np.ascontiguousarray(L[..., 0]) if ch > 1 else L, np.ascontiguousarray(R[..., 0]) if ch > 1 else R, win, D)
            got = m.compute(L, R, win, D, sub_bands=sub_bands)
            assert np.array_equal(got, exp), (n, H, W, ch, win, D)
            for i in range(n):                                  # every "GPU" ends with the full map
                assert np.array_equal(m.device_map(i, H, W).cpu().numpy(), exp), i
    m.close()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("sub_bands", [1, 5])
def test_multi_device_resident_frame_rccl_one_rank(rccl_multi, sub_bands):
    """sadbm_compute_multi_device with one RCCL rank: the frame is already on the GPU, the band is read in place."""
    L, R = noise_pair(17, 131, 402, shift=9)
    for win, D in ((15, 128), (7, 30), (21, 256)):
        exp = oracle.box(L, R, win, D)
        got = rccl_multi.compute_device(_dev(L), _dev(R), win, D, sub_bands=sub_bands)
        assert np.array_equal(got.cpu().numpy(), exp), (win, D)
        assert np.array_equal(rccl_multi.device_map(0, *L.shape).cpu().numpy(), exp)


@pytest.mark.parametrize("n,src", [(2, 0), (3, 2), (5, 1), (8, 7)])
@pytest.mark.parametrize("force_copy", [False, True])
def test_multi_device_resident_frame_virtual_devices(n, src, force_copy):
    """The device-resident form on 2-8 "devices" that are all GPU 0: the source entry reads its band in place, the others pull band
    + halo rows with peer copies (MULTI_FORCE_PEER_COPY makes them do so although the device is the same), pitched ROI views,
    BGR frames, H not divisible by n, bands shorter than the window, two calls in a row without a host wait in between."""
    import torch
    m = oa.Multi(n, devices=[0] * n, multi_flags=oa.MULTI_COPY_GATHER | (oa.MULTI_FORCE_PEER_COPY if force_copy else 0))
    for (H, W, ch) in ((97, 150, 1), (11, 90, 1), (64, 257, 3)):
        L, R = noise_pair(H + n, H, W, shift=4, ch=ch)
        L2, R2 = noise_pair(H + n + 1, H, W, shift=7, ch=ch)
        big = torch.zeros((H + 6, W + 10) + ((ch,) if ch > 1 else ()), dtype=torch.uint8, device="cuda")
        big[3:3 + H, 5:5 + W] = _dev(L)                         # an ROI view: pitch > row bytes (2-D peer copies)
        for win, D, sb in ((9, 40, 1), (15, 128, 3)):
            p0 = lambda a: np.ascontiguousarray(a[..., 0]) if ch > 1 else a
            got = m.compute_device(big[3:3 + H, 5:5 + W], _dev(R), win, D, sub_bands=sb, src_index=src)
            got2 = m.compute_device(_dev(L2), _dev(R2), win, D, sub_bands=sb, src_index=src)     # queued behind the first call
            assert np.array_equal(got.cpu().numpy(), oracle.box(p0(L), p0(R), win, D)), (n, src, H, W, ch, win, D)
            exp2 = oracle.box(p0(L2), p0(R2), win, D)
            assert np.array_equal(got2.cpu().numpy(), exp2), (n, src, H, W, ch, win, D, "second call")
            for i in range(n):                                  # every "GPU" ends with the full map of the last call
                assert np.array_equal(m.device_map(i, H, W).cpu().numpy(), exp2), i
    # the host form on the same handle afterwards (shares the staging buffers and the streams)
    L, R = noise_pair(5, 50, 120, shift=3)
    assert np.array_equal(m.compute(L, R, 7, 30), oracle.box(L, R, 7, 30))
    m.close()


def test_multi_device_rejects_foreign_inputs_and_bad_index(rccl_multi):
    L, R = noise_pair(1, 20, 40, shift=2)
    with pytest.raises(oa.SadbmError) as e:
        rccl_multi.compute_device(_dev(L), _dev(R), 7, 30, src_index=3)
    assert e.value.code == -4


@pytest.mark.parametrize("n", [1, 3])
def test_multi_batch_round_robin(rccl_multi, n):
    m = rccl_multi if n == 1 else oa.Multi(n, devices=[0] * n, multi_flags=oa.MULTI_COPY_GATHER)
    pairs = [oracle.rds(1000 + i, 310, 95, 5 + i, 20 + i, 0) for i in range(7)]
    outs = m.compute_batch([p[0] for p in pairs], [p[1] for p in pairs], 7, 64)
    for i, (L, R) in enumerate(pairs):
        assert np.array_equal(outs[i], oracle.box(L, R, 7, 64)), i
    L3 = [np.ascontiguousarray(np.stack([p[0], p[1], p[0]], -1)) for p in pairs]      # BGR: channel 0 gathered on the host
    R3 = [np.ascontiguousarray(np.stack([p[1], p[0], p[1]], -1)) for p in pairs]
    outs = m.compute_batch(L3, R3, 7, 64)
    for i, (L, R) in enumerate(pairs):
        assert np.array_equal(outs[i], oracle.box(L, R, 7, 64)), ("bgr", i)
    if n != 1:
        m.close()


def test_multi_rejects_duplicate_devices_under_rccl():
    with pytest.raises(oa.SadbmError) as e:
        oa.Multi(2, devices=[0, 0])
    assert e.value.code == -4


def test_two_ranks_sharing_the_gpu_rehearse_the_piecewise_exchange():
    """Two processes (gloo, both on GPU 0) run disparity_row_bands with 1/2/4/7 sub-bands, gathered to all and to rank 0
    (tools/gloo_bands_check.py): every variant gives the same map.  Device tensors under a non-RCCL backend travel
    through host memory -- gloo's own device path hung on some piece sizes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(root, "tools", "gloo_bands_check.py")], cwd=root, env=env, capture_output=True,
                       text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("done") == 2, r.stdout[-2000:]
