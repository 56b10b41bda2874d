"""sdsl-lite_amd — MI355X (gfx950) batched rank/select + wavelet-tree query engine.

Drop-in for the rank/select/wt/count hot path of xxsds/sdsl-lite.  The product is
`lib/libsdsl_hip.so` (hand-written HIP kernels behind the C ABI of include/sdsl_hip.h); this
package is the Python host mirror used by the tests and bench.py.  The directory name contains a
hyphen, so import it with `importlib.import_module("sdsl-lite_amd")`.
"""
from . import capi, dist  # noqa: F401
from .engine import *  # noqa: F401,F403
from .engine import (bit_vector, rank_support_v5, select_support_mcl, rrr_vector, sd_vector, wt_huff, csa_wt, count,  # noqa: F401
                     set_timing, last_kernel_ms, set_random_bits, rnd_positions, rnd_positions_device, mt_checkpoints, density_bits,
                     english_text, english_text_repetitive, set_option, last_phases, device_group, device_scratch_bytes, fused_geometry)
