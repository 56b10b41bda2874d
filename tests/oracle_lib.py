"""ctypes access to the CPU checkers (test infrastructure only):

  * `Oracle`  — oracle/_build/libsdsl_oracle.so, the C restatement of the reference algorithms;
  * `Ref`     — oracle/_ref/libsdsl_ref.so, the REAL sdsl-lite headers compiled by oracle/Makefile
                (present wherever /root/reference was available at build time; optional).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libsdsl_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsdsl_ref.so")

_vp, _u64, _i32, _u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32


class OrcBuf(C.Structure):
    _fields_ = [("p", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


def build_oracle() -> str:
    """(Re)builds the C restatement if needed; gcc only, a second or two."""
    src = os.path.join(ORACLE_DIR, "oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ORACLE_DIR, "oracle.h"))):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "_build/libsdsl_oracle.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return ORACLE_SO


def _p(a) -> int:
    return a.ctypes.data


def _u64arr(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.uint64)


def _u8arr(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.uint8)


class _Lib:
    def __init__(self, path, sigs):
        self.L = C.CDLL(path)
        for name, (res, args) in sigs.items():
            fn = getattr(self.L, name)
            fn.restype = res
            fn.argtypes = args


_ORC_SIGS = {
    "orc_sel": (_u32, [_u64, _u32]),
    "orc_hi": (_u32, [_u64]),
    "orc_set_random_bits": (None, [_vp, _u64, _u64]),
    "orc_mt19937_64_fill": (None, [_vp, _u64, _u64]),
    "orc_buf_free": (None, [C.POINTER(OrcBuf)]),
    "orc_rank_v5_build": (_vp, [_vp, _u64, C.c_int]),
    "orc_rank_v5_free": (None, [_vp]),
    "orc_rank_v5_rank": (_u64, [_vp, _u64]),
    "orc_rank_v5_batch": (None, [_vp, _vp, _u64, _vp]),
    "orc_rank_v5_batch_mt": (None, [_vp, _vp, _u64, _vp, C.c_int]),
    "orc_rank_v5_serialize": (C.c_size_t, [_vp, C.POINTER(OrcBuf)]),
    "orc_select_mcl_build": (_vp, [_vp, _u64, C.c_int]),
    "orc_select_mcl_free": (None, [_vp]),
    "orc_select_mcl_arg_cnt": (_u64, [_vp]),
    "orc_select_mcl_batch": (None, [_vp, _vp, _u64, _vp]),
    "orc_select_mcl_serialize": (C.c_size_t, [_vp, C.POINTER(OrcBuf)]),
    "orc_rrr_build": (_vp, [_vp, _u64]),
    "orc_rrr_free": (None, [_vp]),
    "orc_rrr_size": (_u64, [_vp]),
    "orc_rrr_rank_batch": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "orc_rrr_select_batch": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "orc_rrr_access": (C.c_int, [_vp, _u64]),
    "orc_rrr_serialize": (C.c_size_t, [_vp, C.POINTER(OrcBuf)]),
    "orc_wt_build": (_vp, [_vp, _u64]),
    "orc_wt_free": (None, [_vp]),
    "orc_wt_size": (_u64, [_vp]),
    "orc_wt_sigma": (_u64, [_vp]),
    "orc_wt_bv_size": (_u64, [_vp]),
    "orc_wt_bv_words": (_vp, [_vp]),
    "orc_wt_rank": (_u64, [_vp, _u64, C.c_uint8]),
    "orc_wt_access": (C.c_uint8, [_vp, _u64]),
    "orc_wt_inverse_select": (_u64, [_vp, _u64, _vp]),
    "orc_wt_select": (_u64, [_vp, _u64, C.c_uint8]),
    "orc_wt_rank_batch": (None, [_vp, _vp, _vp, _u64, _vp]),
    "orc_wt_serialize": (C.c_size_t, [_vp, C.c_int, C.POINTER(OrcBuf)]),
    "orc_wt_code_lengths": (None, [_vp, _vp]),
    "orc_sd_build": (_vp, [_vp, _u64]),
    "orc_sd_build_from_positions": (_vp, [_vp, _u64]),
    "orc_sd_free": (None, [_vp]),
    "orc_sd_size": (_u64, [_vp]),
    "orc_sd_ones": (_u64, [_vp]),
    "orc_sd_wl": (_u32, [_vp]),
    "orc_sd_access": (C.c_int, [_vp, _u64]),
    "orc_sd_rank": (_u64, [_vp, _u64, C.c_int]),
    "orc_sd_select": (_u64, [_vp, _u64, C.c_int]),
    "orc_sd_serialize": (C.c_size_t, [_vp, C.POINTER(OrcBuf)]),
    "orc_csa_build": (_vp, [_vp, _u64]),
    "orc_csa_build_from_bwt": (_vp, [_vp, _u64]),
    "orc_csa_build_ex": (_vp, [_vp, _u64, _u64, _u64]),
    "orc_csa_lf": (_u64, [_vp, _u64]),
    "orc_csa_psi": (_u64, [_vp, _u64]),
    "orc_csa_sa": (_u64, [_vp, _u64]),
    "orc_csa_isa": (_u64, [_vp, _u64]),
    "orc_csa_extract": (_u64, [_vp, _u64, _u64, _vp]),
    "orc_csa_locate": (_u64, [_vp, _vp, _u64, _vp, _u64]),
    "orc_csa_free": (None, [_vp]),
    "orc_csa_size": (_u64, [_vp]),
    "orc_csa_sigma": (_u64, [_vp]),
    "orc_csa_bwt": (_vp, [_vp]),
    "orc_csa_wt": (_vp, [_vp]),
    "orc_csa_alphabet": (None, [_vp, _vp, _vp]),
    "orc_csa_backward_search_char": (_u64, [_vp, _u64, _u64, C.c_uint8, _vp, _vp]),
    "orc_csa_count": (_u64, [_vp, _vp, _u64]),
    "orc_csa_interval": (_u64, [_vp, _vp, _u64, _vp, _vp]),
    "orc_csa_count_batch": (None, [_vp, _vp, _u32, _u64, _vp]),
    "orc_csa_serialize_alphabet": (C.c_size_t, [_vp, C.POINTER(OrcBuf)]),
}

_oracle = None


def oracle() -> "_Lib":
    global _oracle
    if _oracle is None:
        _oracle = _Lib(build_oracle(), _ORC_SIGS)
    return _oracle


def _take(buf: OrcBuf) -> bytes:
    data = C.string_at(buf.p, buf.len) if buf.len else b""
    oracle().L.orc_buf_free(C.byref(buf))
    return data


def set_random_bits(n_bits: int, seed: int) -> np.ndarray:
    w = np.zeros((n_bits + 63) // 64 + 1, dtype=np.uint64)  # +1 = SDSL's padding word
    oracle().L.orc_set_random_bits(_p(w), n_bits, seed)
    return w[:-1] if n_bits else w[:0]


def mt19937_64(n: int, seed: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint64)
    oracle().L.orc_mt19937_64_fill(_p(out), n, seed)
    return out


def padded(words, n_bits) -> np.ndarray:
    """SDSL allocates one padding word behind the data (memory_management.hpp:892-918); the
    restatement reads it exactly where SDSL does (rank(size()) with size%64==0)."""
    nw = (n_bits + 63) // 64
    w = np.zeros(nw + 2, dtype=np.uint64)
    w[:nw] = _u64arr(words)[:nw]
    return w


class OBitVector:
    """oracle: bit_vector + rank_support_v5<0/1> + select_support_mcl<0/1>"""

    def __init__(self, words, n_bits):
        L = oracle().L
        self.n_bits = n_bits
        self.words = padded(words, n_bits)
        self.rank_h = [L.orc_rank_v5_build(_p(self.words), n_bits, b) for b in (0, 1)]
        self.sel_h = [L.orc_select_mcl_build(_p(self.words), n_bits, b) for b in (0, 1)]

    def rank(self, idx, bit=1):
        idx = _u64arr(idx)
        out = np.empty(idx.size, dtype=np.uint64)
        oracle().L.orc_rank_v5_batch(self.rank_h[bit], _p(idx), idx.size, _p(out))
        return out

    def select(self, i, bit=1):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint64)
        oracle().L.orc_select_mcl_batch(self.sel_h[bit], _p(i), i.size, _p(out))
        return out

    def arg_cnt(self, bit=1):
        return oracle().L.orc_select_mcl_arg_cnt(self.sel_h[bit])

    def serialize_rank(self, bit=1) -> bytes:
        b = OrcBuf()
        oracle().L.orc_rank_v5_serialize(self.rank_h[bit], C.byref(b))
        return _take(b)

    def serialize_select(self, bit=1) -> bytes:
        b = OrcBuf()
        oracle().L.orc_select_mcl_serialize(self.sel_h[bit], C.byref(b))
        return _take(b)

    def __del__(self):
        try:
            L = oracle().L
            for h in self.rank_h:
                L.orc_rank_v5_free(h)
            for h in self.sel_h:
                L.orc_select_mcl_free(h)
        except Exception:
            pass


class ORrr:
    def __init__(self, words, n_bits):
        self.words = padded(words, n_bits)
        if n_bits % 64:
            self.words[(n_bits - 1) // 64] &= np.uint64((1 << (n_bits % 64)) - 1)
        self.n_bits = n_bits
        self.h = oracle().L.orc_rrr_build(_p(self.words), n_bits)

    def rank(self, i, bit=1):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint64)
        oracle().L.orc_rrr_rank_batch(self.h, bit, _p(i), i.size, _p(out))
        return out

    def select(self, i, bit=1):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint64)
        oracle().L.orc_rrr_select_batch(self.h, bit, _p(i), i.size, _p(out))
        return out

    def access(self, i):
        L = oracle().L
        return np.array([L.orc_rrr_access(self.h, int(x)) for x in _u64arr(i)], dtype=np.uint8)

    def serialize(self) -> bytes:
        b = OrcBuf()
        oracle().L.orc_rrr_serialize(self.h, C.byref(b))
        return _take(b)

    def __del__(self):
        try:
            oracle().L.orc_rrr_free(self.h)
        except Exception:
            pass


class OWt:
    def __init__(self, text: bytes | np.ndarray, handle=None, owner=None):
        self._owner = owner
        if handle is not None:
            self.h = handle
            return
        self.text = _u8arr(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text)
        self.h = oracle().L.orc_wt_build(_p(self.text) if self.text.size else None, self.text.size)

    def size(self):
        return oracle().L.orc_wt_size(self.h)

    def sigma(self):
        return oracle().L.orc_wt_sigma(self.h)

    def bv_size(self):
        return oracle().L.orc_wt_bv_size(self.h)

    def bv_words(self) -> np.ndarray:
        n = (self.bv_size() + 63) // 64
        ptr = oracle().L.orc_wt_bv_words(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n,)).copy() if n else np.zeros(0, np.uint64)

    def rank(self, i, c):
        i, c = _u64arr(i), _u8arr(c)
        out = np.empty(i.size, dtype=np.uint64)
        oracle().L.orc_wt_rank_batch(self.h, _p(i), _p(c), i.size, _p(out))
        return out

    def access(self, i):
        L = oracle().L
        return np.array([L.orc_wt_access(self.h, int(x)) for x in _u64arr(i)], dtype=np.uint8)

    def inverse_select(self, i):
        L = oracle().L
        r, cs = [], []
        cc = C.c_uint8(0)
        for x in _u64arr(i):
            r.append(L.orc_wt_inverse_select(self.h, int(x), C.byref(cc)))
            cs.append(cc.value)
        return np.array(r, dtype=np.uint64), np.array(cs, dtype=np.uint8)

    def select(self, i, c):
        L = oracle().L
        return np.array([L.orc_wt_select(self.h, int(a), int(b)) for a, b in zip(_u64arr(i), _u8arr(c))],
                        dtype=np.uint64)

    def code_lengths(self):
        out = np.zeros(256, dtype=np.uint8)
        oracle().L.orc_wt_code_lengths(self.h, _p(out))
        return out

    def serialize(self, select_is_mcl=1) -> bytes:
        b = OrcBuf()
        oracle().L.orc_wt_serialize(self.h, select_is_mcl, C.byref(b))
        return _take(b)

    def __del__(self):
        try:
            if self._owner is None:
                oracle().L.orc_wt_free(self.h)
        except Exception:
            pass


class OSd:
    """oracle sd_vector<>"""

    def __init__(self, words=None, n_bits=None, positions=None):
        L = oracle().L
        if positions is not None:
            p = _u64arr(positions)
            self.h = L.orc_sd_build_from_positions(_p(p) if p.size else None, p.size)
        else:
            w = padded(_u64arr(words), n_bits)
            self.h = L.orc_sd_build(_p(w), n_bits)

    def size(self):
        return oracle().L.orc_sd_size(self.h)

    def ones(self):
        return oracle().L.orc_sd_ones(self.h)

    def wl(self):
        return oracle().L.orc_sd_wl(self.h)

    def rank(self, idx, bit=1):
        L = oracle().L
        return np.array([L.orc_sd_rank(self.h, int(i), bit) for i in idx], dtype=np.uint64)

    def select(self, idx, bit=1):
        L = oracle().L
        return np.array([L.orc_sd_select(self.h, int(i), bit) for i in idx], dtype=np.uint64)

    def access(self, idx):
        L = oracle().L
        return np.array([L.orc_sd_access(self.h, int(i)) for i in idx], dtype=np.uint8)

    def serialize(self) -> bytes:
        b = OrcBuf()
        oracle().L.orc_sd_serialize(self.h, C.byref(b))
        return _take(b)

    def __del__(self):
        try:
            oracle().L.orc_sd_free(self.h)
        except Exception:
            pass


class OCsa:
    def __init__(self, text: bytes | None = None, bwt: np.ndarray | None = None, sa_dens: int = 32,
                 isa_dens: int = 64):
        L = oracle().L
        if bwt is not None:
            b = _u8arr(bwt)
            self.h = L.orc_csa_build_from_bwt(_p(b), b.size)
        else:
            t = _u8arr(np.frombuffer(text, dtype=np.uint8))
            self.h = L.orc_csa_build_ex(_p(t) if t.size else None, t.size, sa_dens, isa_dens)

    def _each(self, fn, idx):
        f = getattr(oracle().L, fn)
        return np.array([f(self.h, int(i)) for i in idx], dtype=np.uint64)

    def sa(self, idx):
        return self._each("orc_csa_sa", idx)

    def isa(self, idx):
        return self._each("orc_csa_isa", idx)

    def lf(self, idx):
        return self._each("orc_csa_lf", idx)

    def psi(self, idx):
        return self._each("orc_csa_psi", idx)

    def extract(self, begin: int, end: int) -> bytes:
        out = np.empty(end - begin + 1, dtype=np.uint8)
        oracle().L.orc_csa_extract(self.h, begin, end, _p(out))
        return out.tobytes()

    def locate(self, pat: bytes) -> np.ndarray:
        p = _u8arr(np.frombuffer(pat, dtype=np.uint8))
        n = oracle().L.orc_csa_locate(self.h, _p(p) if p.size else None, p.size, None, 0)
        out = np.empty(n, dtype=np.uint64)
        if n:
            oracle().L.orc_csa_locate(self.h, _p(p), p.size, _p(out), n)
        return out

    def size(self):
        return oracle().L.orc_csa_size(self.h)

    def sigma(self):
        return oracle().L.orc_csa_sigma(self.h)

    def bwt(self) -> np.ndarray:
        n = self.size()
        ptr = oracle().L.orc_csa_bwt(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def wt(self) -> OWt:
        return OWt(None, handle=oracle().L.orc_csa_wt(self.h), owner=self)

    def alphabet(self):
        c2c = np.zeros(256, dtype=np.uint8)
        Cc = np.zeros(257, dtype=np.uint64)
        oracle().L.orc_csa_alphabet(self.h, _p(c2c), _p(Cc))
        return c2c, Cc

    def count(self, pat: bytes) -> int:
        p = _u8arr(np.frombuffer(pat, dtype=np.uint8))
        return oracle().L.orc_csa_count(self.h, _p(p) if p.size else None, p.size)

    def interval(self, pat: bytes):
        p = _u8arr(np.frombuffer(pat, dtype=np.uint8))
        l, r = C.c_uint64(0), C.c_uint64(0)
        oracle().L.orc_csa_interval(self.h, _p(p) if p.size else None, p.size, C.byref(l), C.byref(r))
        return l.value, r.value

    def backward_search(self, l, r, c):
        lo, ro = C.c_uint64(0), C.c_uint64(0)
        oracle().L.orc_csa_backward_search_char(self.h, int(l), int(r), int(c), C.byref(lo), C.byref(ro))
        return lo.value, ro.value

    def count_batch(self, pats: np.ndarray, m: int):
        pats = _u8arr(pats)
        n = pats.size // m if m else 0
        out = np.empty(n, dtype=np.uint64)
        oracle().L.orc_csa_count_batch(self.h, _p(pats), m, n, _p(out))
        return out

    def serialize_alphabet(self) -> bytes:
        b = OrcBuf()
        oracle().L.orc_csa_serialize_alphabet(self.h, C.byref(b))
        return _take(b)

    def __del__(self):
        try:
            oracle().L.orc_csa_free(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# the real SDSL (optional)
# ---------------------------------------------------------------------------------------------
_REF_SIGS = {
    "ref_free": (None, [_vp]),
    "ref_bv_create": (_vp, [_vp, _u64]),
    "ref_bv_create_rank": (_vp, [_vp, _u64]),
    "ref_bv_destroy": (None, [_vp]),
    "ref_bv_rank": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "ref_bv_rank_mt": (None, [_vp, C.c_int, _vp, _u64, _vp, C.c_int]),
    "ref_bv_rank_mt_timed": (C.c_double, [_vp, C.c_int, _vp, _u64, _vp, C.c_int, C.c_int]),
    "ref_bv_rank_v": (None, [_vp, _vp, _u64, _vp]),
    "ref_bv_select": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "ref_bv_serialize": (None, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_rrr_create": (_vp, [_vp, _u64]),
    "ref_rrr_destroy": (None, [_vp]),
    "ref_rrr_rank": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "ref_rrr_select": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "ref_rrr_access": (None, [_vp, _vp, _u64, _vp]),
    "ref_rrr_get_int": (None, [_vp, _vp, C.c_uint32, _u64, _vp]),
    "ref_rrr_serialize": (None, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_wt_create": (_vp, [_vp, _u64]),
    "ref_wt_destroy": (None, [_vp]),
    "ref_wt_size": (_u64, [_vp]),
    "ref_wt_sigma": (_u64, [_vp]),
    "ref_wt_bv_size": (_u64, [_vp]),
    "ref_wt_rank": (None, [_vp, _vp, _vp, _u64, _vp]),
    "ref_wt_access": (None, [_vp, _vp, _u64, _vp]),
    "ref_wt_inverse_select": (None, [_vp, _vp, _u64, _vp, _vp]),
    "ref_wt_select": (None, [_vp, _vp, _vp, _u64, _vp]),
    "ref_wt_serialize": (None, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_csa_create": (_vp, [_vp, _u64, C.c_int]),
    "ref_csa_load": (_vp, [_vp, _u64]),
    "ref_csa_destroy": (None, [_vp]),
    "ref_csa_size": (_u64, [_vp]),
    "ref_csa_sigma": (_u64, [_vp]),
    "ref_csa_bwt": (None, [_vp, _vp]),
    "ref_csa_alphabet": (None, [_vp, _vp, _vp]),
    "ref_csa_count": (None, [_vp, _vp, _u32, _u64, _vp]),
    "ref_csa_count_ragged": (None, [_vp, _vp, _vp, _u64, _vp]),
    "ref_csa_interval": (None, [_vp, _vp, _u32, _u64, _vp, _vp]),
    "ref_csa_backward_search": (None, [_vp, _vp, _vp, _vp, _u64, _vp, _vp]),
    "ref_csa_serialize": (None, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_csa_wt_rank": (None, [_vp, _vp, _vp, _u64, _vp]),
    "ref_csa_access": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "ref_csa_extract": (_u64, [_vp, _u64, _u64, _vp]),
    "ref_csa_locate": (_u64, [_vp, _vp, _u64, _vp, _u64]),
    "ref_wt_rrr_serialize": (None, [_vp, _u64, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_csa_rrr_serialize": (C.c_int, [_vp, _u64, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_sd_create": (_vp, [_vp, _u64]),
    "ref_sd_create_from_positions": (_vp, [_vp, _u64]),
    "ref_sd_destroy": (None, [_vp]),
    "ref_sd_size": (_u64, [_vp]),
    "ref_sd_query": (None, [_vp, C.c_int, _vp, _u64, _vp]),
    "ref_sd_serialize": (None, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_bv_pattern": (None, [_vp, _u64, C.c_int, C.c_int, _vp, _u64, _vp]),
    "ref_wt_default_serialize": (None, [_vp, _u64, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_csa_default_serialize": (C.c_int, [_vp, _u64, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_wt_shape_serialize": (None, [_vp, _u64, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_csa_blcd_serialize": (C.c_int, [_vp, _u64, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_set_random_bits": (None, [_vp, _u64, C.c_int]),
    "ref_density_bits": (None, [_vp, _u64, _u64, C.c_uint32]),
    "ref_sibling_serialize": (None, [_vp, _u64, C.c_int, C.POINTER(_vp), C.POINTER(_u64)]),
    "ref_bits_sel": (_u32, [_u64, _u32]),
    "ref_bits_hi": (_u32, [_u64]),
}

_ref = None


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref() -> "_Lib":
    global _ref
    if _ref is None:
        _ref = _Lib(REF_SO, _REF_SIGS)
    return _ref


def _ref_bytes(fn, *args) -> bytes:
    p, n = _vp(None), _u64(0)
    fn(*args, C.byref(p), C.byref(n))
    data = C.string_at(p, n.value) if n.value else b""
    ref().L.ref_free(p)
    return data


def ref_wt_rrr_bytes(text: bytes) -> bytes:
    """wt_huff<rrr_vector<63>>::serialize of the real library"""
    t = _u8arr(np.frombuffer(text, dtype=np.uint8))
    return _ref_bytes(ref().L.ref_wt_rrr_serialize, _p(t) if t.size else None, t.size)


def ref_bv_pattern(words, n_bits, pat: int, which: int, q) -> np.ndarray:
    """real SDSL two-bit pattern supports: pat 0..3 = <10,2> <01,2> <00,2> <11,2>; which 0 = rank_support_v5,
    1 = rank_support_v, 2 = select_support_mcl"""
    w = padded(_u64arr(words), n_bits)
    q = _u64arr(q)
    out = np.empty(q.size, dtype=np.uint64)
    ref().L.ref_bv_pattern(_p(w), n_bits, pat, which, _p(q), q.size, _p(out))
    return out


def pattern_bits(bits: np.ndarray, pat: int) -> np.ndarray:
    """occurrence vector of a two-bit pattern over a 0/1 array (rank_support.hpp:160-284 restated on plain arrays):
    bit i is set iff (x[i-1], x[i]) is the pattern; the bit in front of position 0 is 0 for 10 and 11, 1 for 01 and 00"""
    x = bits.astype(np.uint8)
    prev = np.concatenate([[1 if pat in (1, 2) else 0], x[:-1]]).astype(np.uint8) if x.size else x
    want_prev, want_cur = [(1, 0), (0, 1), (0, 0), (1, 1)][pat]
    return ((prev == want_prev) & (x == want_cur)).astype(np.uint8)


_ref_r15 = None


def ref_sibling_bytes(words, n_bits: int, kind: int) -> bytes:
    """serialize() of bit_vector_il<512> (0), generic rrr_vector<15> (1), bit_vector_il<64> (2), generic
    rrr_vector<15, int_vector<>, 8> (3), rrr_vector<31> (4), rrr_vector<62, int_vector<>, 16> (5), and the rrr_vector<15>
    specialisation of rrr_vector_15.hpp (6)"""
    w = padded(words, n_bits)
    if kind == 6:  # a library of its own: the specialisation and the generic template cannot share one shared object
        global _ref_r15
        if _ref_r15 is None:
            _ref_r15 = C.CDLL(os.path.join(os.path.dirname(REF_SO), "libsdsl_ref_r15.so"), mode=os.RTLD_LOCAL)
            _ref_r15.ref_rrr15_spec_serialize.restype = None
            _ref_r15.ref_rrr15_spec_serialize.argtypes = [_vp, _u64, C.POINTER(_vp), C.POINTER(_u64)]
        return _ref_bytes(_ref_r15.ref_rrr15_spec_serialize, _p(w), n_bits)
    return _ref_bytes(ref().L.ref_sibling_serialize, _p(w), n_bits, kind)


def ref_wt_default_bytes(text: bytes) -> bytes:
    """wt_huff<>::serialize (SDSL's default arguments: rank_support_v, select_support_mcl) of the real library"""
    t = _u8arr(np.frombuffer(text, dtype=np.uint8))
    return _ref_bytes(ref().L.ref_wt_default_serialize, _p(t) if t.size else None, t.size)


def ref_csa_default_bytes(text: bytes) -> bytes:
    """csa_wt<>::serialize of the real library (default wavelet tree, densities 32 / 64)"""
    t = _u8arr(np.frombuffer(text, dtype=np.uint8))
    p, n = _vp(None), _u64(0)
    if ref().L.ref_csa_default_serialize(_p(t), t.size, C.byref(p), C.byref(n)):
        raise ValueError("sdsl::construct_im threw")
    data = C.string_at(p, n.value)
    ref().L.ref_free(p)
    return data


def ref_wt_shape_bytes(text: bytes, shape: int, flavour: int) -> bytes:
    """wt_blcd (shape 1) / wt_hutu (shape 2) ::serialize of the real library; flavour 0 = default template arguments,
    1 = <bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>"""
    t = _u8arr(np.frombuffer(text, dtype=np.uint8))
    return _ref_bytes(ref().L.ref_wt_shape_serialize, _p(t) if t.size else None, t.size, shape, flavour)


def ref_csa_blcd_bytes(text: bytes) -> bytes:
    """csa_wt<wt_blcd<bit_vector, rank_support_v5<>, scan, scan>, 32, 64>::serialize of the real library"""
    t = _u8arr(np.frombuffer(text, dtype=np.uint8))
    p, n = _vp(None), _u64(0)
    if ref().L.ref_csa_blcd_serialize(_p(t), t.size, C.byref(p), C.byref(n)):
        raise ValueError("sdsl::construct_im threw")
    data = C.string_at(p, n.value)
    ref().L.ref_free(p)
    return data


def ref_csa_rrr_bytes(text: bytes) -> bytes:
    """csa_wt<wt_huff<rrr_vector<63>>, 32, 64>::serialize of the real library"""
    t = _u8arr(np.frombuffer(text, dtype=np.uint8))
    p, n = _vp(None), _u64(0)
    if ref().L.ref_csa_rrr_serialize(_p(t), t.size, C.byref(p), C.byref(n)):
        raise ValueError("sdsl::construct_im threw")
    data = C.string_at(p, n.value)
    ref().L.ref_free(p)
    return data


class RBitVector:
    def __init__(self, words, n_bits):
        self.words = padded(words, n_bits)
        self.h = ref().L.ref_bv_create(_p(self.words), n_bits)

    def rank(self, idx, bit=1):
        idx = _u64arr(idx)
        out = np.empty(idx.size, dtype=np.uint64)
        ref().L.ref_bv_rank(self.h, bit, _p(idx), idx.size, _p(out))
        return out

    def rank_v(self, idx):
        idx = _u64arr(idx)
        out = np.empty(idx.size, dtype=np.uint64)
        ref().L.ref_bv_rank_v(self.h, _p(idx), idx.size, _p(out))
        return out

    def select(self, i, bit=1):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint64)
        ref().L.ref_bv_select(self.h, bit, _p(i), i.size, _p(out))
        return out

    def serialize(self, which) -> bytes:
        return _ref_bytes(ref().L.ref_bv_serialize, self.h, which)

    def __del__(self):
        try:
            ref().L.ref_bv_destroy(self.h)
        except Exception:
            pass


class RRrr:
    def __init__(self, words, n_bits):
        self.words = padded(words, n_bits)
        self.h = ref().L.ref_rrr_create(_p(self.words), n_bits)

    def rank(self, i, bit=1):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint64)
        ref().L.ref_rrr_rank(self.h, bit, _p(i), i.size, _p(out))
        return out

    def select(self, i, bit=1):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint64)
        ref().L.ref_rrr_select(self.h, bit, _p(i), i.size, _p(out))
        return out

    def access(self, i):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint8)
        ref().L.ref_rrr_access(self.h, _p(i), i.size, _p(out))
        return out

    def get_int(self, i, length):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint64)
        ref().L.ref_rrr_get_int(self.h, _p(i), length, i.size, _p(out))
        return out

    def serialize(self) -> bytes:
        return _ref_bytes(ref().L.ref_rrr_serialize, self.h)

    def __del__(self):
        try:
            ref().L.ref_rrr_destroy(self.h)
        except Exception:
            pass


class RWt:
    def __init__(self, text):
        self.text = _u8arr(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text)
        self.h = ref().L.ref_wt_create(_p(self.text) if self.text.size else None, self.text.size)

    def size(self):
        return ref().L.ref_wt_size(self.h)

    def sigma(self):
        return ref().L.ref_wt_sigma(self.h)

    def bv_size(self):
        return ref().L.ref_wt_bv_size(self.h)

    def rank(self, i, c):
        i, c = _u64arr(i), _u8arr(c)
        out = np.empty(i.size, dtype=np.uint64)
        ref().L.ref_wt_rank(self.h, _p(i), _p(c), i.size, _p(out))
        return out

    def access(self, i):
        i = _u64arr(i)
        out = np.empty(i.size, dtype=np.uint8)
        ref().L.ref_wt_access(self.h, _p(i), i.size, _p(out))
        return out

    def inverse_select(self, i):
        i = _u64arr(i)
        r = np.empty(i.size, dtype=np.uint64)
        c = np.empty(i.size, dtype=np.uint8)
        ref().L.ref_wt_inverse_select(self.h, _p(i), i.size, _p(r), _p(c))
        return r, c

    def select(self, i, c):
        i, c = _u64arr(i), _u8arr(c)
        out = np.empty(i.size, dtype=np.uint64)
        ref().L.ref_wt_select(self.h, _p(i), _p(c), i.size, _p(out))
        return out

    def serialize(self, select_is_mcl=1) -> bytes:
        return _ref_bytes(ref().L.ref_wt_serialize, self.h, select_is_mcl)

    def __del__(self):
        try:
            ref().L.ref_wt_destroy(self.h)
        except Exception:
            pass


class RSd:
    """the real sd_vector<> with its rank / select supports"""

    def __init__(self, words=None, n_bits=None, positions=None):
        L = ref().L
        if positions is not None:
            p = _u64arr(positions)
            self.h = L.ref_sd_create_from_positions(_p(p) if p.size else None, p.size)
        else:
            w = padded(_u64arr(words), n_bits)
            self.h = L.ref_sd_create(_p(w), n_bits)

    def size(self):
        return ref().L.ref_sd_size(self.h)

    def _q(self, what, q):
        q = _u64arr(q)
        out = np.empty(q.size, dtype=np.uint64)
        ref().L.ref_sd_query(self.h, what, _p(q), q.size, _p(out))
        return out

    def rank(self, idx, bit=1):
        return self._q(1 if bit else 0, idx)

    def select(self, i, bit=1):
        return self._q(3 if bit else 2, i)

    def access(self, idx):
        return self._q(4, idx).astype(np.uint8)

    def serialize(self) -> bytes:
        return _ref_bytes(ref().L.ref_sd_serialize, self.h)

    def __del__(self):
        try:
            ref().L.ref_sd_destroy(self.h)
        except Exception:
            pass


class RCsa:
    def __init__(self, text: bytes | None = None, also_fm_huff=False, sdsl_bytes=None):
        """from a text (sdsl::construct_im) or from the serialised bytes of csa_wt<wt_huff<bit_vector,
        rank_support_v5<>>> (sa / isa density 32 / 64), which the real library then loads"""
        if sdsl_bytes is not None:
            b = _u8arr(np.frombuffer(sdsl_bytes, dtype=np.uint8) if isinstance(sdsl_bytes, (bytes, bytearray)) else sdsl_bytes)
            self.h = ref().L.ref_csa_load(_p(b), b.size)
            if not self.h:
                raise ValueError("sdsl::csa_wt::load failed")
            return
        t = _u8arr(np.frombuffer(text, dtype=np.uint8))
        self.h = ref().L.ref_csa_create(_p(t) if t.size else None, t.size, 1 if also_fm_huff else 0)
        if not self.h:
            raise ValueError("sdsl::construct_im threw (text contains a 0 byte?)")

    def size(self):
        return ref().L.ref_csa_size(self.h)

    def sigma(self):
        return ref().L.ref_csa_sigma(self.h)

    def bwt(self):
        out = np.empty(self.size(), dtype=np.uint8)
        ref().L.ref_csa_bwt(self.h, _p(out))
        return out

    def alphabet(self):
        c2c = np.zeros(256, dtype=np.uint8)
        Cc = np.zeros(257, dtype=np.uint64)
        ref().L.ref_csa_alphabet(self.h, _p(c2c), _p(Cc))
        return c2c, Cc

    def count_batch(self, pats, m):
        pats = _u8arr(pats)
        n = pats.size // m if m else 0
        out = np.empty(n, dtype=np.uint64)
        ref().L.ref_csa_count(self.h, _p(pats), m, n, _p(out))
        return out

    def count(self, pat: bytes) -> int:
        p = _u8arr(np.frombuffer(pat, dtype=np.uint8))
        offs = np.array([0, p.size], dtype=np.uint64)
        out = np.empty(1, dtype=np.uint64)
        pp = p if p.size else np.zeros(1, np.uint8)
        ref().L.ref_csa_count_ragged(self.h, _p(pp), _p(offs), 1, _p(out))
        return int(out[0])

    def interval_batch(self, pats, m):
        pats = _u8arr(pats)
        n = pats.size // m
        l = np.empty(n, dtype=np.uint64)
        r = np.empty(n, dtype=np.uint64)
        ref().L.ref_csa_interval(self.h, _p(pats), m, n, _p(l), _p(r))
        return l, r

    def backward_search(self, l, r, c):
        l, r, c = _u64arr(l), _u64arr(r), _u8arr(c)
        lo = np.empty(l.size, dtype=np.uint64)
        ro = np.empty(l.size, dtype=np.uint64)
        ref().L.ref_csa_backward_search(self.h, _p(l), _p(r), _p(c), l.size, _p(lo), _p(ro))
        return lo, ro

    def wt_rank(self, i, c):
        i, c = _u64arr(i), _u8arr(c)
        out = np.empty(i.size, dtype=np.uint64)
        ref().L.ref_csa_wt_rank(self.h, _p(i), _p(c), i.size, _p(out))
        return out

    def _access(self, what, idx):
        idx = _u64arr(idx)
        out = np.empty(idx.size, dtype=np.uint64)
        ref().L.ref_csa_access(self.h, what, _p(idx), idx.size, _p(out))
        return out

    def sa(self, idx):
        return self._access(0, idx)

    def isa(self, idx):
        return self._access(1, idx)

    def lf(self, idx):
        return self._access(2, idx)

    def psi(self, idx):
        return self._access(3, idx)

    def extract(self, begin: int, end: int) -> bytes:
        out = np.empty(end - begin + 1, dtype=np.uint8)
        ref().L.ref_csa_extract(self.h, begin, end, _p(out))
        return out.tobytes()

    def locate(self, pat: bytes) -> np.ndarray:
        p = _u8arr(np.frombuffer(pat, dtype=np.uint8))
        n = ref().L.ref_csa_locate(self.h, _p(p), p.size, None, 0)
        out = np.empty(max(n, 1), dtype=np.uint64)
        ref().L.ref_csa_locate(self.h, _p(p), p.size, _p(out), n)
        return out[:n]

    def serialize(self, which=0) -> bytes:
        return _ref_bytes(ref().L.ref_csa_serialize, self.h, which)

    def __del__(self):
        try:
            ref().L.ref_csa_destroy(self.h)
        except Exception:
            pass
