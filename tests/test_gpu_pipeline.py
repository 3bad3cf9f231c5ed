"""Streaming pipeline (sadbm_pipeline_*): overlapped H2D / matcher / D2H over pinned slots.
Every frame must equal the oracle; tickets, slot reuse and the BUSY rule are checked."""
import numpy as np
import pytest

import opencv_amd as oa
import oracle
from helpers import noise_pair

pytestmark = pytest.mark.gpu


def test_pipeline_stream_of_frames_matches_oracle():
    H, W, win, D = 120, 333, 9, 64
    frames = [oracle.rds(200 + i, W, H, 3 + i, 30 + i, 0) for i in range(7)]
    exp = [oracle.box(L, R, win, D) for L, R in frames]
    pipe = oa.Pipeline(H, W, 1, win, D, depth=3)
    tickets = []
    for i, (L, R) in enumerate(frames):
        if i >= 3:                                          # slot reuse: drain the ticket that owns the slot first
            assert np.array_equal(pipe.wait(tickets[i - 3]), exp[i - 3])
        tickets.append(pipe.submit(L, R))
    assert tickets == list(range(7))
    for i in range(4, 7):
        assert np.array_equal(pipe.wait(tickets[i], copy=True), exp[i]), i
    # waiting twice is fine while the slot has not been reused; a recycled ticket is refused
    assert np.array_equal(pipe.wait(tickets[6]), exp[6])
    with pytest.raises(oa.SadbmError) as e:
        pipe.wait(tickets[0])
    assert e.value.code == -7
    pipe.close()


def test_pipeline_zero_copy_inputs_bgr_and_busy():
    H, W, win, D = 64, 200, 15, 128
    L, R = oracle.rds(9, W, H, 5, 70, 0)
    L3 = np.stack([L, R, L], -1)
    R3 = np.stack([R, L, R], -1)
    exp = oracle.box(L, R, win, D)
    pipe = oa.Pipeline(H, W, 3, win, D, depth=2)
    for _ in range(2):
        li, ri = pipe.inputs()
        li[...] = L3
        ri[...] = R3
        pipe.submit()
    with pytest.raises(oa.SadbmError) as e:                 # both slots in flight / unwaited
        pipe.submit(L3, R3)
    assert e.value.code == -7
    assert np.array_equal(pipe.wait(0), exp) and np.array_equal(pipe.wait(1), exp)
    t = pipe.submit(L3, R3)
    assert t == 2 and np.array_equal(pipe.wait(t), exp)
    with pytest.raises(oa.SadbmError):
        pipe.submit(L, R)                                   # channel count differs from the pipeline's


@pytest.mark.gpu
def test_host_api_is_reentrant_across_threads():
    """the reference keeps its state in file-scope globals (SAD.cpp:9-11) and is not re-entrant; here every host
    thread gets its own lazily created context, so concurrent getDisparity() calls with different parameters do not disturb
    each other.  8 threads x 6 calls against the oracle."""
    import threading
    jobs = [(7, 30, (97, 211)), (9, 64, (120, 300)), (15, 128, (80, 517)), (5, 16, (60, 90)), (21, 200, (70, 400)), (4, 40, (33, 257)),
            (11, 100, (64, 640)), (13, 48, (50, 129))]
    data = []
    for n, (win, D, (H, W)) in enumerate(jobs):
        L, R = noise_pair(4000 + n, H, W, shift=3 + n)
        data.append((win, D, L, R, oracle.box(L, R, win, D)))
    errors = []

    def worker(k):
        try:
            for rep in range(6):
                win, D, L, R, exp = data[(k + rep) % len(data)]
                got = oa.getDisparity(L, R, winSize=win, DSR=D)
                if not np.array_equal(got, exp):
                    errors.append((k, rep, win, D))
        except Exception as e:      # noqa: BLE001 -- reported below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
