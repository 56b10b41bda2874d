"""ctypes binding of include/sdsl_hip.h (the C ABI of libsdsl_hip.so).

The library is loaded lazily and loudly: if the HIP extension has not been built, or no
gfx950 device is visible when a structure is created, callers get an exception — there is no
CPU fallback anywhere in this package (the CPU restatement lives under oracle/ and is test
infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDSL_HIP_LIB") or os.path.join(HERE, "lib", "libsdsl_hip.so")  # (SDSL_HIP_LIB: another build of the same library, tools/ab_fused.sh)
HEADER_PATH = os.path.join(HERE, "..", "include", "sdsl_hip.h")

OK = 0
ERR_INVALID, ERR_NOMEM, ERR_HIP, ERR_FORMAT, ERR_NO_DEVICE, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
NPOS = 0xFFFFFFFFFFFFFFFF
BV_SELECT1, BV_SELECT0 = 1, 2
WT_RRR63 = 1
WT_BLCD = 2
WT_HUTU = 4
LAYOUT_BV_SCAN, LAYOUT_BV_MCL, LAYOUT_RRR63, LAYOUT_BV_DEFAULT = 0, 1, 2, 3
SIBLING_IL, SIBLING_RRR15 = 0, 1


def SIBLING_RRR(t_bs: int, t_k: int = 32) -> int:
    """generic rrr_vector<t_bs, int_vector<>, t_k> (SDSL_HIP_SIBLING_RRR)"""
    return 2 | (t_bs << 8) | (t_k << 16)

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p

# name -> (restype, argtypes); pointers to batch arrays are passed as raw addresses (c_void_p)
# because they may be host or device memory.
SIGNATURES = {
    "sdsl_hip_allocated_bytes": (C.c_uint64, []),
    "sdsl_hip_last_error": (C.c_char_p, []),
    "sdsl_hip_version": (C.c_char_p, []),
    "sdsl_hip_device_count": (C.c_int32, []),
    "sdsl_hip_set_option": (C.c_int32, [C.c_char_p, C.c_int64]),
    "sdsl_hip_limit": (C.c_uint64, [C.c_char_p]),
    "sdsl_hip_last_phases": (C.c_int32, [C.c_char_p, C.c_size_t]),
    "sdsl_hip_bv_layout_info": (C.c_int32, [_vp, C.POINTER(C.c_uint64)]),
    "sdsl_hip_util_set_random_bits": (C.c_int32, [_vp, C.c_uint64, C.c_uint64]),
    "sdsl_hip_util_rnd_positions": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "sdsl_hip_util_mt_checkpoints": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "sdsl_hip_util_density_bits": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, C.c_uint64, C.c_uint64]),
    "sdsl_hip_util_english_text": (C.c_int32, [_vp, C.c_uint64, C.c_uint64]),
    "sdsl_hip_util_english_text_repetitive": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, C.c_uint32]),
    "sdsl_hip_util_rnd_positions_device": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _vp, C.c_int32,
                                                       _vp]),
    "sdsl_hip_group_create": (C.c_int32, [_vp, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_group_destroy": (C.c_int32, [_vp]),
    "sdsl_hip_group_size": (C.c_int32, [_vp]),
    "sdsl_hip_group_device": (C.c_int32, [_vp, C.c_int32]),
    "sdsl_hip_group_loopback": (C.c_int32, [_vp, C.c_uint64, C.POINTER(C.c_float)]),
    "sdsl_hip_group_bv_replicate": (C.c_int32, [_vp, _vp, C.POINTER(_vp)]),
    "sdsl_hip_group_bv_rank_batch": (C.c_int32, [_vp, C.POINTER(_vp), C.c_int32, _vp, C.c_uint64, _vp, C.c_int32]),
    "sdsl_hip_group_bv_select_batch": (C.c_int32, [_vp, C.POINTER(_vp), C.c_int32, _vp, C.c_uint64, _vp, C.c_int32]),
    "sdsl_hip_group_fm_create_from_text": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, C.POINTER(_vp)]),
    "sdsl_hip_group_fm_count_batch": (C.c_int32, [_vp, C.POINTER(_vp), _vp, C.c_uint32, C.c_uint64, _vp, C.c_int32]),
    "sdsl_hip_bv_create": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(_vp)]),
    "sdsl_hip_bv_create_from_sdsl": (C.c_int32, [_vp, C.c_size_t, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(_vp)]),
    "sdsl_hip_bv_add_select": (C.c_int32, [_vp, C.c_uint32]),
    "sdsl_hip_bv_create_pattern": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.POINTER(_vp)]),
    "sdsl_hip_bv_serialize": (C.c_int32, [_vp, C.c_int32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sdsl_hip_bv_destroy": (C.c_int32, [_vp]),
    "sdsl_hip_bv_release_scratch": (C.c_int32, [_vp]),
    "sdsl_hip_bv_reserve_capture_scratch": (C.c_int32, [_vp, C.c_uint64]),
    "sdsl_hip_rrr_reserve_capture_scratch": (C.c_int32, [_vp, C.c_uint64]),
    "sdsl_hip_device_scratch_bytes": (C.c_uint64, [C.c_int32]),
    "sdsl_hip_bv_query_one": (C.c_int32, [_vp, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "sdsl_hip_bv_size": (C.c_uint64, [_vp]),
    "sdsl_hip_bv_ones": (C.c_uint64, [_vp]),
    "sdsl_hip_bv_device_bytes": (C.c_uint64, [_vp]),
    "sdsl_hip_bv_rank_batch": (C.c_int32, [_vp, C.c_int32, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_bv_gather_probe": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_bv_select_batch": (C.c_int32, [_vp, C.c_int32, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_bv_access_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_bv_export_words": (C.c_int32, [_vp, _vp, _vp]),
    "sdsl_hip_rrr_create": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_rrr_create_from_sdsl": (C.c_int32, [_vp, C.c_size_t, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_rrr_create_from_sibling": (C.c_int32, [_vp, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_rrr_serialize": (C.c_int32, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sdsl_hip_rrr_destroy": (C.c_int32, [_vp]),
    "sdsl_hip_rrr_size": (C.c_uint64, [_vp]),
    "sdsl_hip_rrr_ones": (C.c_uint64, [_vp]),
    "sdsl_hip_rrr_device_bytes": (C.c_uint64, [_vp]),
    "sdsl_hip_rrr_rank_batch": (C.c_int32, [_vp, C.c_int32, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_rrr_get_int_batch": (C.c_int32, [_vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp]),
    "sdsl_hip_rrr_select_batch": (C.c_int32, [_vp, C.c_int32, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_rrr_access_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_sd_create": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_sd_create_from_positions": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_sd_create_from_sdsl": (C.c_int32, [_vp, C.c_size_t, C.c_int32, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "sdsl_hip_sd_serialize": (C.c_int32, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sdsl_hip_sd_destroy": (C.c_int32, [_vp]),
    "sdsl_hip_sd_size": (C.c_uint64, [_vp]),
    "sdsl_hip_sd_ones": (C.c_uint64, [_vp]),
    "sdsl_hip_sd_low_width": (C.c_uint32, [_vp]),
    "sdsl_hip_sd_lane_kernels": (C.c_uint32, [_vp]),
    "sdsl_hip_sd_device_bytes": (C.c_uint64, [_vp]),
    "sdsl_hip_sd_rank_batch": (C.c_int32, [_vp, C.c_int32, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_sd_select_batch": (C.c_int32, [_vp, C.c_int32, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_sd_access_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_wt_create": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_wt_create_ex": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(_vp)]),
    "sdsl_hip_wt_create_from_sdsl": (C.c_int32, [_vp, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(_vp),
                                                 C.POINTER(C.c_size_t)]),
    "sdsl_hip_wt_serialize": (C.c_int32, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sdsl_hip_wt_serialize_ex": (C.c_int32, [_vp, C.c_int32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sdsl_hip_wt_destroy": (C.c_int32, [_vp]),
    "sdsl_hip_wt_size": (C.c_uint64, [_vp]),
    "sdsl_hip_wt_sigma": (C.c_uint64, [_vp]),
    "sdsl_hip_wt_bv_size": (C.c_uint64, [_vp]),
    "sdsl_hip_wt_device_bytes": (C.c_uint64, [_vp]),
    "sdsl_hip_wt_code_lengths": (C.c_int32, [_vp, _vp]),
    "sdsl_hip_wt_fused_steps": (C.c_int32, [_vp, _vp]),
    "sdsl_hip_wt_fused_geometry": (None, [_vp, _vp, _vp]),
    "sdsl_hip_wt_release_binary_levels": (C.c_int32, [_vp]),
    "sdsl_hip_wt_rank_batch": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_wt_access_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_wt_inverse_select_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "sdsl_hip_wt_select_batch": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_fm_create_from_bwt": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_fm_create_from_text": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_fm_create_from_bwt_ex": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(_vp)]),
    "sdsl_hip_fm_create_from_text_ex": (C.c_int32, [_vp, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(_vp)]),
    "sdsl_hip_fm_create_from_sdsl": (C.c_int32, [_vp, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "sdsl_hip_fm_serialize": (C.c_int32, [_vp, C.c_uint32, C.c_uint32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sdsl_hip_fm_serialize_ex": (C.c_int32, [_vp, C.c_int32, C.c_uint32, C.c_uint32, _vp, C.c_size_t,
                                             C.POINTER(C.c_size_t)]),
    "sdsl_hip_fm_create_from_sdsl_ex": (C.c_int32, [_vp, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32,
                                                   C.POINTER(_vp)]),
    "sdsl_hip_fm_drop_sa": (C.c_int32, [_vp]),
    "sdsl_hip_fm_drop_sa_ex": (C.c_int32, [_vp, C.c_uint32, C.c_uint32]),
    "sdsl_hip_fm_restore_suffix_array": (C.c_int32, [_vp]),
    "sdsl_hip_fm_set_jump_depth": (C.c_int32, [_vp, C.c_uint32]),
    "sdsl_hip_fm_jump_depth": (C.c_uint32, [_vp]),
    "sdsl_hip_fm_set_kmer_table": (C.c_int32, [_vp, C.c_uint32, C.c_uint64]),
    "sdsl_hip_fm_kmer_table_depth": (C.c_uint32, [_vp]),
    "sdsl_hip_fm_kmer_table_bytes": (C.c_uint64, [_vp]),
    "sdsl_hip_fm_set_footprint": (C.c_int32, [_vp, C.c_uint64]),
    "sdsl_hip_fm_footprint_parts": (None, [_vp, _vp]),
    "sdsl_hip_fm_sampling": (C.c_int32, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "sdsl_hip_fm_sa_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_fm_isa_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_fm_lf_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_fm_psi_batch": (C.c_int32, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_fm_extract_batch": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64,
                                              C.POINTER(C.c_uint64), _vp]),
    "sdsl_hip_fm_sa_range_batch": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64,
                                               C.POINTER(C.c_uint64), _vp]),
    "sdsl_hip_fm_locate_batch": (C.c_int32, [_vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp, C.c_uint64,
                                             C.POINTER(C.c_uint64), _vp]),
    "sdsl_hip_fm_destroy": (C.c_int32, [_vp]),
    "sdsl_hip_fm_size": (C.c_uint64, [_vp]),
    "sdsl_hip_fm_sigma": (C.c_uint64, [_vp]),
    "sdsl_hip_fm_device_bytes": (C.c_uint64, [_vp]),
    "sdsl_hip_fm_wavelet_tree": (_vp, [_vp]),
    "sdsl_hip_fm_alphabet": (C.c_int32, [_vp, _vp, _vp]),
    "sdsl_hip_fm_backward_search_batch": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "sdsl_hip_fm_count_batch": (C.c_int32, [_vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp]),
    "sdsl_hip_fm_count_ragged": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _vp, _vp]),
    "sdsl_hip_fm_interval_batch": (C.c_int32, [_vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp, _vp]),
    "sdsl_hip_set_timing": (C.c_int32, [C.c_int32]),
    "sdsl_hip_last_kernel_ms": (C.c_int32, [C.POINTER(C.c_float)]),
}


class SdslHipError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"sdsl_hip status {status}: {msg}")
        self.status = status


def declared_symbols(header_path: str = HEADER_PATH) -> list[str]:
    """Every function the public header declares (used by the symbol-export test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdsl_hip_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib() -> C.CDLL:
    """Loads libsdsl_hip.so (once).  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python sdsl-lite_amd/build.py` (or __graft_entry__.build()). "
            "This package has no CPU fallback.")
    try:  # share torch's HIP runtime (same SONAME libamdhip64.so.7) when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the pure C ABI
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            continue  # reported by the symbol-export test; calling it raises AttributeError
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status: int) -> None:
    if status != OK:
        raise SdslHipError(status, lib().sdsl_hip_last_error().decode(errors="replace"))
