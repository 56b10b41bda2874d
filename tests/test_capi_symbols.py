"""CPU: the C-ABI library loads without a GPU, exports every function include/sdsl_hip.h declares,
fails loudly (no CPU fallback) when no device is present, and its host-only helpers work."""
import ctypes as C

import numpy as np
import pytest


def test_every_declared_symbol_is_exported(pkg):
    lib = pkg.capi.lib()
    declared = pkg.capi.declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in include/sdsl_hip.h but not exported: {missing}"
    assert sorted(pkg.capi.SIGNATURES) == declared, "capi.SIGNATURES out of sync with the header"


def test_no_cpu_fallback_without_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    w = np.zeros(4, dtype=np.uint64)
    with pytest.raises(pkg.capi.SdslHipError) as e:
        pkg.bit_vector(w, 256)
    assert e.value.status == pkg.capi.ERR_NO_DEVICE
    assert pkg.capi.lib().sdsl_hip_device_count() == 0


def test_set_random_bits_matches_std_mt19937_64(pkg):
    import oracle_lib as ol
    assert np.array_equal(pkg.set_random_bits(1 << 16, 815), ol.set_random_bits(1 << 16, 815))


def test_invalid_arguments(pkg):
    lib = pkg.capi.lib()
    assert lib.sdsl_hip_bv_create(None, 64, 0, 0, None) == pkg.capi.ERR_INVALID
    assert lib.sdsl_hip_bv_rank_batch(None, 1, None, 0, None, None) == pkg.capi.ERR_INVALID
    assert b"" != lib.sdsl_hip_last_error()


def test_fused_geometry_of_the_build(pkg):
    """the form of the fused wavelet-tree lines is a compile-time choice (wt_device.hpp: SDSL_HIP_FUSED_K); the library says which one
    it was built with, and the numbers hang together: four sections of a 128-byte line, relative counts that fit their fields"""
    g = pkg.fused_geometry()
    assert g["levels_per_fetch"] in (3, 4)
    if g["levels_per_fetch"] == 4:  # 16-ary: 46 positions per section, counts of 18 bits relative to superblocks of 1024 lines
        assert g["positions_per_line"] % 4 == 0 and g["positions_per_line"] in (192, 184, 176)
        bits = 16 + (192 - g["positions_per_line"]) // 4
        assert g["lines_per_superblock"] * g["positions_per_line"] < (1 << bits)
    else:                          # 8-ary: 64 positions per section, absolute 32-bit counts
        assert g["positions_per_line"] == 256 and g["lines_per_superblock"] == 0
