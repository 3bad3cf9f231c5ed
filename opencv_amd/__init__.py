"""opencv_amd -- MI355X-native SAD stereo block matching (the hot path of the reference's SAD.cpp).

Host-side mirror of the reference interface (`getDisparity`, `winSize`, `DSR`) over the C ABI of
include/sadbm.h.  There is no CPU fallback: every compute call goes through libsadbm.so (HIP,
gfx950) and raises if the library or a GPU is missing.
"""
from .stereo import (  # noqa: F401
    RAW_LOC, FORCE_GENERIC, KERNEL_GLOBAL, KERNEL_STRIP, KERNEL_QUAD, SadbmError, Context, Pipeline, getDisparity, get_disparity_host, library_path,
    load_library, build_library, compute_device, compute_band_device, compute_batch_device,
    DEFAULT_WIN_SIZE, DEFAULT_DSR, Multi, MULTI_COPY_GATHER, MULTI_FORCE_PEER_COPY, band_plan, compute_device_static, plan, reload_env, thread_release,
)
from .bands import band_ranges, band_input_rows  # noqa: F401
from .multi import shard_pairs, scatter_band, disparity_row_bands  # noqa: F401
from . import filters  # noqa: F401,E402  (GetgaussianKernel, filter2D, highpass, blend: SURVEY 8f f4)
