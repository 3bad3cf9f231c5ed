"""A plain C++ program (tests/cpp/abi_consumer.cpp: g++, include/sadbm.hpp only, no Python, no HIP headers) consumes the C
ABI on the GPU box: getDisparity, the single-process multi-GPU entry points (RCCL with one rank; three "devices" on GPU 0
with the peer-copy gather and pipelined sub-bands), the batch dealer and the streaming pipeline agree with each other and
with the hand-derivable known answers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "opencv_amd", "csrc", "abi_consumer")


@pytest.mark.gpu
def test_cpp_consumer_of_the_c_abi():
    if not os.path.exists(EXE):
        pytest.skip("abi_consumer not built (make -C opencv_amd/csrc)")
    # standalone process: libsadbm.so binds to the ROCm runtime of the image, sadbm_multi_create dlopens /opt/rocm's librccl
    env = dict(os.environ)
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0 and "abi_consumer: ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
