"""Python host mirror of the SDSL concepts served by libsdsl_hip (used by tests/ and bench.py).

Class and method names follow the reference (bit_vector / rank_support_v5 / select_support_mcl /
rrr_vector / wt_huff / csa_wt / count); every query method takes an ARRAY of arguments and
returns an array, on the device when the input is a CUDA(HIP) torch tensor and on the host when
it is a numpy array.  torch is plumbing here: device memory, streams, torch.distributed.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _ptr(x) -> int:
    if _is_tensor(x):
        return x.data_ptr()
    return x.ctypes.data


def _stream_for(x) -> int:
    if _is_tensor(x) and x.is_cuda:
        return torch.cuda.current_stream(x.device).cuda_stream
    return 0


def _as_array(x, dtype, name):
    """Contiguous array of `dtype`: torch tensors stay where they are, everything else -> numpy."""
    if _is_tensor(x):
        want = {np.uint64: (torch.int64, torch.uint64), np.uint8: (torch.uint8,)}[dtype]
        if x.dtype not in want:
            raise TypeError(f"{name}: expected {want}, got {x.dtype}")
        return x.contiguous()
    a = np.ascontiguousarray(x, dtype=dtype)
    return a


def _empty_like(x, n, dtype):
    if _is_tensor(x):
        tdt = {np.uint64: torch.int64, np.uint8: torch.uint8}[dtype]
        return torch.empty(n, dtype=tdt, device=x.device)
    return np.empty(n, dtype=dtype)


def _out_for(x, n, dtype, out):
    """A fresh result array beside x, or the caller's `out` after checking that the library may write n items of
    `dtype` into it (size, dtype, contiguity, same side of the PCIe link as x)."""
    if out is None:
        return _empty_like(x, n, dtype)
    if _is_tensor(out) != _is_tensor(x):
        raise TypeError("out must live where the arguments live (torch tensor with torch tensor, numpy with numpy)")
    if _is_tensor(out):
        want = {np.uint64: (torch.int64, torch.uint64), np.uint8: (torch.uint8,)}[dtype]
        if out.dtype not in want or not out.is_contiguous() or out.numel() < n or out.device != x.device:
            raise ValueError(f"out: need a contiguous {want[0]} tensor of at least {n} elements on {x.device}")
    else:
        if not isinstance(out, np.ndarray) or out.dtype != np.dtype(dtype) or not out.flags.c_contiguous or out.size < n \
                or not out.flags.writeable:
            raise ValueError(f"out: need a writeable C-contiguous {np.dtype(dtype)} array of at least {n} elements")
    return out


class _Handle:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p(None)
        self._own = True

    def close(self):
        if getattr(self, "_h", None) is not None and self._h and self._own:
            getattr(capi.lib(), self._destroy)(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class bit_vector(_Handle):
    """Device-resident sdsl::bit_vector with rank_support_v5<0/1> and select_support_mcl<0/1>.

    words: uint64 array (numpy, or torch int64/uint64 tensor on host or device) = bit_vector::data()
    """
    _destroy = "sdsl_hip_bv_destroy"

    def __init__(self, words=None, n_bits: int | None = None, device: int = 0, select1: bool = True,
                 select0: bool = True, pattern: tuple[int, int] | None = None, sdsl_bytes: bytes | None = None,
                 kind: int | None = None):
        """pattern=(t_b, t_pat_len) as in SDSL's templates, e.g. (10, 2): the handle then holds the occurrence vector
        of that two-bit pattern, and rank(idx, 1) / select(i, 1) answer rank_support_v5<10,2> / select_support_mcl<10,2>.
        sdsl_bytes + kind (capi.SIBLING_IL / capi.SIBLING_RRR15): the serialised bytes of a bit_vector_il<> /
        rrr_vector<15>, decoded to plain bits on the device."""
        super().__init__()
        if sdsl_bytes is not None:
            buf = np.frombuffer(sdsl_bytes, dtype=np.uint8)
            flags = (capi.BV_SELECT1 if select1 else 0) | (capi.BV_SELECT0 if select0 else 0)
            capi.check(capi.lib().sdsl_hip_bv_create_from_sdsl(_ptr(buf), buf.size, kind, device, flags, C.byref(self._h)))
            self.device = device
            return
        w = _as_array(words, np.uint64, "words")
        nw = w.numel() if _is_tensor(w) else w.size
        if n_bits is None:
            n_bits = nw * 64
        if (n_bits + 63) // 64 > nw:
            raise ValueError("words too short for n_bits")
        flags = (capi.BV_SELECT1 if select1 else 0) | (capi.BV_SELECT0 if select0 else 0)
        if pattern is None:
            capi.check(capi.lib().sdsl_hip_bv_create(_ptr(w) if nw else None, n_bits, device, flags, C.byref(self._h)))
        else:
            capi.check(capi.lib().sdsl_hip_bv_create_pattern(_ptr(w) if nw else None, n_bits, device, pattern[0],
                                                             pattern[1], flags, C.byref(self._h)))
        self.device = device

    def size(self) -> int:
        return capi.lib().sdsl_hip_bv_size(self._h)

    __len__ = size

    def ones(self) -> int:
        return capi.lib().sdsl_hip_bv_ones(self._h)

    def device_bytes(self) -> int:
        return capi.lib().sdsl_hip_bv_device_bytes(self._h)

    def layout_info(self) -> dict:
        o = (C.c_uint64 * 4)()
        capi.check(capi.lib().sdsl_hip_bv_layout_info(self._h, o))
        return {"lines_ptr": int(o[0]), "lines_bytes": int(o[1]), "select1_ptr": int(o[2]), "scratch_ptr": int(o[3])}

    def release_scratch(self):
        """Frees the working memory the bucketed batch rank keeps with the handle."""
        capi.check(capi.lib().sdsl_hip_bv_release_scratch(self._h))

    def reserve_capture_scratch(self, max_queries: int):
        """Working memory, owned by this handle, for large batches enqueued while their stream is being captured into a
        graph (sdsl_hip.h: stream capture); 0 releases it."""
        capi.check(capi.lib().sdsl_hip_bv_reserve_capture_scratch(self._h, max_queries))

    def rank(self, idx, bit: int = 1, out=None):
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_bv_rank_batch(self._h, bit, _ptr(idx), n, _ptr(out), _stream_for(idx)))
        return out

    def gather_probe(self, idx, out):
        """measurement aid: the memory-access skeleton of rank() without its arithmetic (device tensors only)"""
        idx = _as_array(idx, np.uint64, "idx")
        capi.check(capi.lib().sdsl_hip_bv_gather_probe(self._h, _ptr(idx), idx.numel(), _ptr(out), _stream_for(idx)))
        return out

    def select(self, i, bit: int = 1, out=None):
        i = _as_array(i, np.uint64, "i")
        n = i.numel() if _is_tensor(i) else i.size
        out = _out_for(i, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_bv_select_batch(self._h, bit, _ptr(i), n, _ptr(out), _stream_for(i)))
        return out

    def access(self, idx, out=None):
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint8, out)
        capi.check(capi.lib().sdsl_hip_bv_access_batch(self._h, _ptr(idx), n, _ptr(out), _stream_for(idx)))
        return out

    __getitem__ = access

    def serialize(self, what: int = 0) -> bytes:
        """SDSL's own bytes: 0 bit_vector, 1/2 rank_support_v5<1>/<0>, 3/4 select_support_mcl<1>/<0>,
        5/6 rank_support_v<1>/<0>"""
        L = capi.lib()
        need = C.c_size_t(0)
        capi.check(L.sdsl_hip_bv_serialize(self._h, what, None, 0, C.byref(need)))
        buf = np.empty(max(1, need.value), dtype=np.uint8)
        capi.check(L.sdsl_hip_bv_serialize(self._h, what, _ptr(buf), need.value, C.byref(need)))
        return buf[: need.value].tobytes()

    def export_words(self) -> np.ndarray:
        out = np.zeros((self.size() + 63) // 64, dtype=np.uint64)
        capi.check(capi.lib().sdsl_hip_bv_export_words(self._h, _ptr(out) if out.size else None, 0))
        return out


class rank_support_v5:
    """rank_support_v5<t_b> view of a device bit_vector (rank_support_v5.hpp:44)."""

    def __init__(self, bv: bit_vector, t_b: int = 1):
        self.m_v, self.t_b = bv, t_b

    def rank(self, idx, out=None):
        return self.m_v.rank(idx, self.t_b, out)

    __call__ = rank

    def size(self):
        return self.m_v.size()


class select_support_mcl:
    """select_support_mcl<t_b> view of a device bit_vector (select_support_mcl.hpp:64)."""

    def __init__(self, bv: bit_vector, t_b: int = 1):
        self.m_v, self.t_b = bv, t_b

    def select(self, i, out=None):
        return self.m_v.select(i, self.t_b, out)

    __call__ = select

    def size(self):
        return self.m_v.size()


def set_timing(enabled: bool) -> None:
    capi.check(capi.lib().sdsl_hip_set_timing(1 if enabled else 0))


def last_kernel_ms() -> float:
    ms = C.c_float(0)
    capi.check(capi.lib().sdsl_hip_last_kernel_ms(C.byref(ms)))
    return float(ms.value)


def set_option(name: str, value: int) -> None:
    """sdsl_hip_set_option, e.g. set_option("rank_sorted", 1)."""
    capi.check(capi.lib().sdsl_hip_set_option(name.encode(), int(value)))


def device_scratch_bytes(device: int = 0) -> int:
    """Bytes of the device's scratch pool for the bucketed batch paths (shared by all handles; release_scratch frees it)."""
    return int(capi.lib().sdsl_hip_device_scratch_bytes(int(device)))


def last_phases() -> dict:
    """Per-pass milliseconds of the most recent bucketed batch rank (needs set_option("trace_phases", 1))."""
    buf = C.create_string_buffer(1024)
    capi.check(capi.lib().sdsl_hip_last_phases(buf, 1024))
    out = {}
    for kv in buf.value.decode().split(";"):
        if "=" in kv:
            k, v = kv.split("=")
            out[k] = float(v)
    return out


def set_random_bits(n_bits: int, seed: int) -> np.ndarray:
    """util::set_random_bits on a fresh bit_vector(n_bits): returns the uint64 words."""
    w = np.zeros((n_bits + 63) // 64, dtype=np.uint64)
    if w.size:
        capi.check(capi.lib().sdsl_hip_util_set_random_bits(_ptr(w), n_bits, seed))
    return w


def rnd_positions(seed: int, count: int, mod: int = 0, add: int = 0) -> np.ndarray:
    """add + std::mt19937_64(seed)() % mod, `count` successive draws (SURVEY.md 8(d) query streams)."""
    out = np.empty(count, dtype=np.uint64)
    if count:
        capi.check(capi.lib().sdsl_hip_util_rnd_positions(seed, count, mod, add, _ptr(out)))
    return out


def mt_checkpoints(seed: int, stride: int, n: int) -> np.ndarray:
    """Generator states of mt19937_64(seed) before draws 0, stride, 2*stride, ...: n rows of 313 words."""
    out = np.empty((n, 313), dtype=np.uint64)
    capi.check(capi.lib().sdsl_hip_util_mt_checkpoints(seed, stride, n, _ptr(out)))
    return out


def rnd_positions_device(seed: int, count: int, mod: int = 0, add: int = 0, device: int = 0, stride: int = 1 << 20):
    """rnd_positions, written straight into device memory (int64 tensor): the host only walks the generator once to save its
    state every `stride` draws (a few MB); the device regenerates the stream from those checkpoints."""
    out = torch.empty(count, dtype=torch.int64, device=torch.device("cuda", device))
    if count:
        n_ck = (count + stride - 1) // stride
        ck = mt_checkpoints(seed, stride, n_ck)
        capi.check(capi.lib().sdsl_hip_util_rnd_positions_device(_ptr(ck), n_ck, stride, count, mod, add, _ptr(out), device,
                                                                 torch.cuda.current_stream(out.device).cuda_stream))
    return out


def density_bits(n_bits: int, seed: int, percent: int, checkpoints: np.ndarray | None = None, stride: int = 0) -> np.ndarray:
    """bit i = (i-th draw of mt19937_64(seed) % 100 < percent): the configs[2] vector, as uint64 words."""
    w = np.zeros((n_bits + 63) // 64, dtype=np.uint64)
    if checkpoints is not None:
        cp = np.ascontiguousarray(checkpoints, dtype=np.uint64).reshape(-1, 313)
        capi.check(capi.lib().sdsl_hip_util_density_bits(_ptr(w), n_bits, seed, percent, _ptr(cp), cp.shape[0], stride))
    else:
        capi.check(capi.lib().sdsl_hip_util_density_bits(_ptr(w), n_bits, seed, percent, None, 0, 0))
    return w


def fused_geometry() -> dict:
    """the form of the fused wavelet-tree lines the library was built with (sdsl_hip_wt_fused_geometry)"""
    lv, pp, sb = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    capi.lib().sdsl_hip_wt_fused_geometry(C.byref(lv), C.byref(pp), C.byref(sb))
    return {"levels_per_fetch": lv.value, "positions_per_line": pp.value, "lines_per_superblock": sb.value}


def english_text(n_bytes: int, seed: int) -> np.ndarray:
    """The English-class stand-in text of the configs[3]/[4] benchmarks (uint8, no zero byte)."""
    out = np.empty(n_bytes, dtype=np.uint8)
    if n_bytes:
        capi.check(capi.lib().sdsl_hip_util_english_text(_ptr(out), n_bytes, seed))
    return out


def english_text_repetitive(n_bytes: int, seed: int, percent: int = 30) -> np.ndarray:
    """english_text with `percent` of its 64 KiB blocks replaced by rotated copies of earlier blocks (duplicated passages, as in a
    real collection): patterns drawn from it keep wide suffix-array intervals for many characters."""
    out = np.empty(n_bytes, dtype=np.uint8)
    if n_bytes:
        capi.check(capi.lib().sdsl_hip_util_english_text_repetitive(_ptr(out), n_bytes, seed, percent))
    return out


def _serialize(fn, handle) -> bytes:
    need = C.c_size_t(0)
    capi.check(fn(handle, None, 0, C.byref(need)))
    buf = np.empty(max(1, need.value), dtype=np.uint8)
    capi.check(fn(handle, _ptr(buf), need.value, C.byref(need)))
    return buf[: need.value].tobytes()


class rrr_vector(_Handle):
    """Device rrr_vector<63, int_vector<>, 32> with its rank/select supports (rrr_vector.hpp:68)."""
    _destroy = "sdsl_hip_rrr_destroy"

    def __init__(self, words=None, n_bits: int | None = None, device: int = 0, sdsl_bytes: bytes | None = None, sibling_kind: int | None = None):
        """sibling_kind (capi.SIBLING_IL / SIBLING_RRR15 / SIBLING_RRR(t_bs, t_k)): sdsl_bytes are the stream of that sibling type; it is
        decoded on the device and kept compressed as rrr records"""
        super().__init__()
        if sdsl_bytes is not None and sibling_kind is not None:
            buf = np.frombuffer(sdsl_bytes, dtype=np.uint8)
            capi.check(capi.lib().sdsl_hip_rrr_create_from_sibling(_ptr(buf), buf.size, sibling_kind, device, C.byref(self._h)))
        elif sdsl_bytes is not None:
            buf = np.frombuffer(sdsl_bytes, dtype=np.uint8)
            capi.check(capi.lib().sdsl_hip_rrr_create_from_sdsl(_ptr(buf), buf.size, device, C.byref(self._h)))
        else:
            w = _as_array(words, np.uint64, "words")
            nw = w.numel() if _is_tensor(w) else w.size
            if n_bits is None:
                n_bits = nw * 64
            if (n_bits + 63) // 64 > nw:
                raise ValueError(f"words too short for n_bits: {nw} words < ceil({n_bits} / 64)")
            capi.check(capi.lib().sdsl_hip_rrr_create(_ptr(w) if nw else None, n_bits, device, C.byref(self._h)))
        self.device = device

    def reserve_capture_scratch(self, max_queries: int):
        capi.check(capi.lib().sdsl_hip_rrr_reserve_capture_scratch(self._h, max_queries))

    def size(self) -> int:
        return capi.lib().sdsl_hip_rrr_size(self._h)

    def ones(self) -> int:
        return capi.lib().sdsl_hip_rrr_ones(self._h)

    def device_bytes(self) -> int:
        return capi.lib().sdsl_hip_rrr_device_bytes(self._h)

    def rank(self, idx, bit: int = 1, out=None):
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_rrr_rank_batch(self._h, bit, _ptr(idx), n, _ptr(out), _stream_for(idx)))
        return out

    def select(self, i, bit: int = 1, out=None):
        i = _as_array(i, np.uint64, "i")
        n = i.numel() if _is_tensor(i) else i.size
        out = _out_for(i, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_rrr_select_batch(self._h, bit, _ptr(i), n, _ptr(out), _stream_for(i)))
        return out

    def access(self, idx, out=None):
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint8, out)
        capi.check(capi.lib().sdsl_hip_rrr_access_batch(self._h, _ptr(idx), n, _ptr(out), _stream_for(idx)))
        return out

    def get_int(self, idx, length: int = 64, out=None):
        """rrr_vector::get_int(idx, len) for an array of idx (rrr_vector.hpp:308-356)."""
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_rrr_get_int_batch(self._h, _ptr(idx), length, n, _ptr(out), _stream_for(idx)))
        return out

    def serialize(self) -> bytes:
        """the bytes sdsl::rrr_vector<63>::serialize writes for the same bit vector"""
        return _serialize(capi.lib().sdsl_hip_rrr_serialize, self._h)

    __getitem__ = access


class sd_vector(_Handle):
    """Device sd_vector<> (Elias-Fano coded sparse bit vector, sd_vector.hpp:134) with rank_support_sd /
    select_support_sd semantics.  From a plain bit vector (words, n_bits), from strictly increasing positions
    (positions, n_bits) or from sd_vector<>::serialize bytes."""
    _destroy = "sdsl_hip_sd_destroy"

    def __init__(self, words=None, n_bits: int | None = None, positions=None, device: int = 0,
                 sdsl_bytes: bytes | None = None):
        super().__init__()
        L = capi.lib()
        self.consumed = None
        if sdsl_bytes is not None:
            buf = np.frombuffer(sdsl_bytes, dtype=np.uint8)
            used = C.c_size_t(0)
            capi.check(L.sdsl_hip_sd_create_from_sdsl(_ptr(buf), buf.size, device, C.byref(self._h), C.byref(used)))
            self.consumed = used.value
        elif positions is not None:
            p = _as_array(positions, np.uint64, "positions")
            m = p.numel() if _is_tensor(p) else p.size
            if n_bits is None:
                raise ValueError("positions need n_bits (the size of the bit vector)")
            capi.check(L.sdsl_hip_sd_create_from_positions(_ptr(p) if m else None, m, n_bits, device, C.byref(self._h)))
        else:
            w = _as_array(words, np.uint64, "words")
            nw = w.numel() if _is_tensor(w) else w.size
            if n_bits is None:
                n_bits = nw * 64
            if (n_bits + 63) // 64 > nw:
                raise ValueError("words too short for n_bits")
            capi.check(L.sdsl_hip_sd_create(_ptr(w) if nw else None, n_bits, device, C.byref(self._h)))
        self.device = device

    def size(self) -> int:
        return capi.lib().sdsl_hip_sd_size(self._h)

    __len__ = size

    def ones(self) -> int:
        return capi.lib().sdsl_hip_sd_ones(self._h)

    def low_width(self) -> int:
        return capi.lib().sdsl_hip_sd_low_width(self._h)

    def lane_kernels(self) -> int:
        """bit 0: rank / access, bit 1: select_0 are answered by the one-lane-per-query kernels (sd.hip)"""
        return capi.lib().sdsl_hip_sd_lane_kernels(self._h)

    def serialize(self) -> bytes:
        """the bytes of sd_vector<>::serialize"""
        return _serialize(capi.lib().sdsl_hip_sd_serialize, self._h)

    def device_bytes(self) -> int:
        return capi.lib().sdsl_hip_sd_device_bytes(self._h)

    def rank(self, idx, bit: int = 1, out=None):
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_sd_rank_batch(self._h, bit, _ptr(idx), n, _ptr(out), _stream_for(idx)))
        return out

    def select(self, i, bit: int = 1, out=None):
        i = _as_array(i, np.uint64, "i")
        n = i.numel() if _is_tensor(i) else i.size
        out = _out_for(i, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_sd_select_batch(self._h, bit, _ptr(i), n, _ptr(out), _stream_for(i)))
        return out

    def access(self, idx, out=None):
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint8, out)
        capi.check(capi.lib().sdsl_hip_sd_access_batch(self._h, _ptr(idx), n, _ptr(out), _stream_for(idx)))
        return out

    __getitem__ = access


def _bytes_arg(x, name):
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    return _as_array(x, np.uint8, name)


class wt_huff(_Handle):
    """Device wt_huff<bit_vector, rank_support_v5<>> over bytes (wt_huff.hpp:62-67, wt_pc.hpp:59)."""
    _destroy = "sdsl_hip_wt_destroy"

    def __init__(self, text=None, device: int = 0, sdsl_bytes: bytes | None = None, select_is_mcl: bool = True,
                 rrr: bool = False, balanced: bool = False, hutu: bool = False, _borrowed=None):
        """rrr=True: wt_huff<rrr_vector<63>> (the bit vector is stored rrr-compressed); for sdsl_bytes it says that
        the stream is of that type, otherwise select_is_mcl tells the two plain flavours apart.  balanced=True builds
        the wt_blcd shape instead of the Huffman shape (streams of any wt_pc byte shape load as they are)."""
        super().__init__()
        self.consumed = None
        if _borrowed is not None:
            self._h = C.c_void_p(_borrowed)
            self._own = False
        elif sdsl_bytes is not None:
            buf = np.frombuffer(sdsl_bytes, dtype=np.uint8)
            used = C.c_size_t(0)
            layout = capi.LAYOUT_RRR63 if rrr else (capi.LAYOUT_BV_MCL if select_is_mcl else capi.LAYOUT_BV_SCAN)
            capi.check(capi.lib().sdsl_hip_wt_create_from_sdsl(_ptr(buf), buf.size, layout, device,
                                                               C.byref(self._h), C.byref(used)))
            self.consumed = used.value
        else:
            t = _bytes_arg(text, "text")
            n = t.numel() if _is_tensor(t) else t.size
            flags = (capi.WT_RRR63 if rrr else 0) | (capi.WT_BLCD if balanced else 0) | (capi.WT_HUTU if hutu else 0)
            capi.check(capi.lib().sdsl_hip_wt_create_ex(_ptr(t) if n else None, n, device, flags, C.byref(self._h)))
        self.device = device

    def size(self) -> int:
        return capi.lib().sdsl_hip_wt_size(self._h)

    def sigma(self) -> int:
        return capi.lib().sdsl_hip_wt_sigma(self._h)

    def bv_size(self) -> int:
        return capi.lib().sdsl_hip_wt_bv_size(self._h)

    def device_bytes(self) -> int:
        return capi.lib().sdsl_hip_wt_device_bytes(self._h)

    def release_binary_levels(self):
        """keep only the fused lines (rank / access / inverse_select / select walk them); serialize rebuilds SDSL's levels for the call"""
        capi.check(capi.lib().sdsl_hip_wt_release_binary_levels(self._h))

    def serialize(self, layout: int = 0) -> bytes:
        """the bytes of wt_huff<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>> (layout 0),
        wt_huff<bit_vector, rank_support_v5<>> with mcl selects (capi.LAYOUT_BV_MCL) or wt_huff<> with SDSL's default
        arguments (capi.LAYOUT_BV_DEFAULT); an rrr tree always writes wt_huff<rrr_vector<63>>"""
        L = capi.lib()
        need = C.c_size_t(0)
        capi.check(L.sdsl_hip_wt_serialize_ex(self._h, layout, None, 0, C.byref(need)))
        buf = np.empty(max(1, need.value), dtype=np.uint8)
        capi.check(L.sdsl_hip_wt_serialize_ex(self._h, layout, _ptr(buf), need.value, C.byref(need)))
        return buf[: need.value].tobytes()

    def code_lengths(self) -> np.ndarray:
        out = np.zeros(256, dtype=np.uint8)
        capi.check(capi.lib().sdsl_hip_wt_code_lengths(self._h, _ptr(out)))
        return out

    def fused_steps(self) -> np.ndarray:
        """line fetches a rank / access / select of each symbol costs on the fused layout (zeros without one)"""
        out = np.zeros(256, dtype=np.uint8)
        capi.check(capi.lib().sdsl_hip_wt_fused_steps(self._h, _ptr(out)))
        return out

    def rank(self, i, c, out=None):
        i = _as_array(i, np.uint64, "i")
        c = _as_array(c, np.uint8, "c")
        n = i.numel() if _is_tensor(i) else i.size
        out = _out_for(i, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_wt_rank_batch(self._h, _ptr(i), _ptr(c), n, _ptr(out), _stream_for(i)))
        return out

    def access(self, i, out=None):
        i = _as_array(i, np.uint64, "i")
        n = i.numel() if _is_tensor(i) else i.size
        out = _out_for(i, n, np.uint8, out)
        capi.check(capi.lib().sdsl_hip_wt_access_batch(self._h, _ptr(i), n, _ptr(out), _stream_for(i)))
        return out

    __getitem__ = access

    def inverse_select(self, i):
        i = _as_array(i, np.uint64, "i")
        n = i.numel() if _is_tensor(i) else i.size
        r = _empty_like(i, n, np.uint64)
        c = _empty_like(i, n, np.uint8)
        capi.check(capi.lib().sdsl_hip_wt_inverse_select_batch(self._h, _ptr(i), n, _ptr(r), _ptr(c), _stream_for(i)))
        return r, c

    def select(self, i, c, out=None):
        i = _as_array(i, np.uint64, "i")
        c = _as_array(c, np.uint8, "c")
        n = i.numel() if _is_tensor(i) else i.size
        out = _out_for(i, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_wt_select_batch(self._h, _ptr(i), _ptr(c), n, _ptr(out), _stream_for(i)))
        return out


class csa_wt(_Handle):
    """Device csa_wt<wt_huff<bit_vector, rank_support_v5<>>> restricted to backward_search/count
    (csa_wt.hpp:56, suffix_array_algorithm.hpp:167-248,464-471)."""
    _destroy = "sdsl_hip_fm_destroy"

    def __init__(self, text=None, bwt=None, device: int = 0, sdsl_bytes: bytes | None = None,
                 select_is_mcl: bool = True, rrr: bool = False, sa_dens: int = 0, isa_dens: int = 0,
                 balanced: bool = False, hutu: bool = False):
        """rrr=True: csa_wt<wt_huff<rrr_vector<63>>> (compressed FM-index).  sa_dens / isa_dens: the template arguments
        of the serialised type (needed to keep its SA / ISA samples for sa / isa / locate / extract)."""
        super().__init__()
        L = capi.lib()
        flags = (capi.WT_RRR63 if rrr else 0) | (capi.WT_BLCD if balanced else 0) | (capi.WT_HUTU if hutu else 0)
        if sdsl_bytes is not None:
            buf = np.frombuffer(sdsl_bytes, dtype=np.uint8)
            layout = capi.LAYOUT_RRR63 if rrr else (capi.LAYOUT_BV_MCL if select_is_mcl else capi.LAYOUT_BV_SCAN)
            capi.check(L.sdsl_hip_fm_create_from_sdsl_ex(_ptr(buf), buf.size, layout, sa_dens, isa_dens, device,
                                                         C.byref(self._h)))
        elif bwt is not None:
            b = _bytes_arg(bwt, "bwt")
            n = b.numel() if _is_tensor(b) else b.size
            capi.check(L.sdsl_hip_fm_create_from_bwt_ex(_ptr(b) if n else None, n, device, flags, C.byref(self._h)))
        else:
            t = _bytes_arg(text, "text")
            n = t.numel() if _is_tensor(t) else t.size
            capi.check(L.sdsl_hip_fm_create_from_text_ex(_ptr(t) if n else None, n, device, flags, C.byref(self._h)))
        self.device = device
        self._wt_ref = None

    @property
    def wavelet_tree(self):
        """csa.wavelet_tree (csa_wt.hpp:130): a view of the tree inside this index.  The view holds the index alive;
        the index only remembers the view weakly (no reference cycle: dropping the last user reference frees the
        HBM at once), and close() detaches it."""
        import weakref
        wt = self._wt_ref() if self._wt_ref is not None else None
        if wt is None:
            if not self._h:
                raise ValueError("csa_wt is closed")
            wt = wt_huff(_borrowed=capi.lib().sdsl_hip_fm_wavelet_tree(self._h), device=self.device)
            wt._keepalive = self
            self._wt_ref = weakref.ref(wt)
        return wt

    def close(self):
        wt = self._wt_ref() if getattr(self, "_wt_ref", None) is not None else None
        if wt is not None:
            wt._h = C.c_void_p(None)  # the borrowed pointer dies with the index
            wt._keepalive = None
        super().close()

    def size(self) -> int:
        return capi.lib().sdsl_hip_fm_size(self._h)

    def sigma(self) -> int:
        return capi.lib().sdsl_hip_fm_sigma(self._h)

    def device_bytes(self) -> int:
        return capi.lib().sdsl_hip_fm_device_bytes(self._h)

    def serialize(self, sa_dens: int, isa_dens: int, layout: int = 0) -> bytes:
        """bytes of csa_wt<wt_huff<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>,
        sa_dens, isa_dens>::serialize — only for an index created from text (the suffix array is needed)"""
        need = C.c_size_t(0)
        L = capi.lib()
        capi.check(L.sdsl_hip_fm_serialize_ex(self._h, layout, sa_dens, isa_dens, None, 0, C.byref(need)))
        buf = np.empty(max(1, need.value), dtype=np.uint8)
        capi.check(L.sdsl_hip_fm_serialize_ex(self._h, layout, sa_dens, isa_dens, _ptr(buf), need.value, C.byref(need)))
        return buf[: need.value].tobytes()

    def drop_sa(self, sa_dens: int = 0, isa_dens: int = 0):
        """release the whole suffix array and the text, keep samples (densities of the caller's csa_wt type; 0 = what the index holds,
        else SDSL's defaults 32 / 64)"""
        capi.check(capi.lib().sdsl_hip_fm_drop_sa_ex(self._h, sa_dens, isa_dens))

    def restore_suffix_array(self):
        """an index loaded from an SDSL stream with its densities gets its text, whole suffix array and k-mer table back"""
        capi.check(capi.lib().sdsl_hip_fm_restore_suffix_array(self._h))

    def jump_depth(self) -> int:
        """characters of a pattern answered by the k-mer interval table instead of LF steps"""
        return capi.lib().sdsl_hip_fm_jump_depth(self._h)

    def set_jump_depth(self, k: int):
        capi.check(capi.lib().sdsl_hip_fm_set_jump_depth(self._h, k))

    def set_kmer_table(self, k_max: int, budget_bytes: int):
        """(re)build the k-mer hash table of count() with the deepest k <= k_max that fits budget_bytes; k_max 0 releases it"""
        capi.check(capi.lib().sdsl_hip_fm_set_kmer_table(self._h, k_max, budget_bytes))

    def set_footprint(self, max_bytes: int):
        """give HBM back until device_bytes() <= max_bytes (binary tree levels first, then suffix array and text -> SDSL's default
        samples with the k-mer table as deep as the budget still allows); answers never change"""
        capi.check(capi.lib().sdsl_hip_fm_set_footprint(self._h, int(max_bytes)))

    def footprint_parts(self) -> dict:
        p = (C.c_uint64 * 8)()
        capi.lib().sdsl_hip_fm_footprint_parts(self._h, p)
        names = ("wt_binary_levels", "wt_fused_lines", "suffix_array", "text", "sa_isa_samples", "kmer_table", "jump_table", "tables")
        return {k: int(v) for k, v in zip(names, p)}

    def kmer_table_depth(self) -> int:
        return capi.lib().sdsl_hip_fm_kmer_table_depth(self._h)

    def kmer_table_bytes(self) -> int:
        return capi.lib().sdsl_hip_fm_kmer_table_bytes(self._h)

    def alphabet(self):
        c2c = np.zeros(256, dtype=np.uint8)
        Cc = np.zeros(257, dtype=np.uint64)
        capi.check(capi.lib().sdsl_hip_fm_alphabet(self._h, _ptr(c2c), _ptr(Cc)))
        return c2c, Cc

    def backward_search(self, l, r, c):
        l = _as_array(l, np.uint64, "l")
        r = _as_array(r, np.uint64, "r")
        c = _as_array(c, np.uint8, "c")
        n = l.numel() if _is_tensor(l) else l.size
        lo, ro = _empty_like(l, n, np.uint64), _empty_like(l, n, np.uint64)
        capi.check(capi.lib().sdsl_hip_fm_backward_search_batch(self._h, _ptr(l), _ptr(r), _ptr(c), n, _ptr(lo),
                                                                _ptr(ro), _stream_for(l)))
        return lo, ro

    def count(self, patterns, m: int, out=None):
        """patterns: n*m bytes (fixed length m) -> uint64[n]"""
        p = _bytes_arg(patterns, "patterns")
        total = p.numel() if _is_tensor(p) else p.size
        n = total // m if m else 0
        out = _out_for(p, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_fm_count_batch(self._h, _ptr(p) if total else None, m, n, _ptr(out),
                                                      _stream_for(p)))
        return out

    def interval(self, patterns, m: int):
        p = _bytes_arg(patterns, "patterns")
        total = p.numel() if _is_tensor(p) else p.size
        n = total // m if m else 0
        l, r = _empty_like(p, n, np.uint64), _empty_like(p, n, np.uint64)
        capi.check(capi.lib().sdsl_hip_fm_interval_batch(self._h, _ptr(p) if total else None, m, n, _ptr(l), _ptr(r),
                                                         _stream_for(p)))
        return l, r

    def sampling(self):
        """(sa_dens, isa_dens, has_full_sa)"""
        a, b, f = C.c_uint32(0), C.c_uint32(0), C.c_int32(0)
        capi.check(capi.lib().sdsl_hip_fm_sampling(self._h, C.byref(a), C.byref(b), C.byref(f)))
        return a.value, b.value, bool(f.value)

    def _simple(self, fn, idx):
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _empty_like(idx, n, np.uint64)
        capi.check(getattr(capi.lib(), fn)(self._h, _ptr(idx) if n else None, n, _ptr(out) if n else None,
                                           _stream_for(idx)))
        return out

    def sa(self, idx):
        """csa[i] (csa_wt.hpp:363-381)"""
        return self._simple("sdsl_hip_fm_sa_batch", idx)

    def isa(self, idx):
        """csa.isa[i] (suffix_array_helper.hpp:519-537)"""
        return self._simple("sdsl_hip_fm_isa_batch", idx)

    def lf(self, idx):
        """csa.lf[i]"""
        return self._simple("sdsl_hip_fm_lf_batch", idx)

    def psi(self, idx):
        """csa.psi[i]"""
        return self._simple("sdsl_hip_fm_psi_batch", idx)

    def _ragged(self, call, like, n, elem_dtype):
        total = C.c_uint64(0)
        offs = _empty_like(like, n + 1, np.uint64)
        capi.check(call(_ptr(offs), None, 0, C.byref(total)))
        out = _empty_like(like, total.value, elem_dtype)
        if total.value:
            capi.check(call(None, _ptr(out), total.value, C.byref(total)))
        return offs, out

    def extract(self, begin, end):
        """text[begin[q]..end[q]] (inclusive) for every q -> (offsets[n+1], bytes)  (suffix_array_algorithm.hpp:578-600)"""
        b = _as_array(begin, np.uint64, "begin")
        e = _as_array(end, np.uint64, "end")
        n = b.numel() if _is_tensor(b) else b.size
        L = capi.lib()
        return self._ragged(lambda o, t, cap, tot: L.sdsl_hip_fm_extract_batch(
            self._h, _ptr(b) if n else None, _ptr(e) if n else None, n, o, t, cap, tot, _stream_for(b)), b, n, np.uint8)

    def sa_range(self, l, r):
        """csa[l[q]..r[q]] for every SA interval -> (offsets[n+1], positions)"""
        l = _as_array(l, np.uint64, "l")
        r = _as_array(r, np.uint64, "r")
        n = l.numel() if _is_tensor(l) else l.size
        L = capi.lib()
        return self._ragged(lambda o, t, cap, tot: L.sdsl_hip_fm_sa_range_batch(
            self._h, _ptr(l) if n else None, _ptr(r) if n else None, n, o, t, cap, tot, _stream_for(l)), l, n, np.uint64)

    def locate(self, patterns, m: int):
        """all occurrences of n fixed-length patterns, SA order within a pattern -> (offsets[n+1], positions)
        (suffix_array_algorithm.hpp:505-523)"""
        l, r = self.interval(patterns, m)
        return self.sa_range(l, r)

    def count_ragged(self, pats: list[bytes]):
        offs = np.zeros(len(pats) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in pats])
        blob = np.frombuffer(b"".join(pats) + b"\0", dtype=np.uint8)  # never empty
        out = np.empty(len(pats), dtype=np.uint64)
        capi.check(capi.lib().sdsl_hip_fm_count_ragged(self._h, _ptr(blob), _ptr(offs), len(pats), _ptr(out), 0))
        return out


def count(csa: csa_wt, patterns, m: int, out=None):
    """sdsl::count(csa, begin, end) over a batch of fixed-length patterns."""
    return csa.count(patterns, m, out)


class device_group(_Handle):
    """Several GPUs of one node driven by ONE process (SURVEY.md 8(e), C ABI sdsl_hip_group_*): the index is replicated with
    one RCCL broadcast per buffer, a batch owned by the root device is scattered, answered by the single-GPU kernels and
    gathered, in `chunks` pipelined pieces."""
    _destroy = "sdsl_hip_group_destroy"

    def __init__(self, devices):
        super().__init__()
        self.devices = list(devices)
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        capi.check(capi.lib().sdsl_hip_group_create(arr, len(self.devices), C.byref(self._h)))

    def __len__(self):
        return len(self.devices)

    def replicate(self, bv: bit_vector):
        """[bv, replica on devices[1], ...]: the replicas are bit_vector objects that own their handles."""
        n = len(self.devices)
        reps = (C.c_void_p * n)()
        capi.check(capi.lib().sdsl_hip_group_bv_replicate(self._h, bv._h, reps))
        out = [bv]
        for r in range(1, n):
            o = bit_vector.__new__(bit_vector)
            _Handle.__init__(o)
            o._h = C.c_void_p(reps[r])
            o.device = self.devices[r]
            out.append(o)
        return out

    def _handles(self, objs):
        return (C.c_void_p * len(objs))(*[o._h.value if isinstance(o._h, C.c_void_p) else o._h for o in objs])

    def rank(self, replicas, idx, bit: int = 1, out=None, chunks: int = 8):
        """The whole batch lives with the root (host memory, or devices[0]); answers come back in the caller's order."""
        idx = _as_array(idx, np.uint64, "idx")
        n = idx.numel() if _is_tensor(idx) else idx.size
        out = _out_for(idx, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_group_bv_rank_batch(self._h, self._handles(replicas), bit, _ptr(idx), n, _ptr(out), chunks))
        return out

    def select(self, replicas, i, bit: int = 1, out=None, chunks: int = 8):
        i = _as_array(i, np.uint64, "i")
        n = i.numel() if _is_tensor(i) else i.size
        out = _out_for(i, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_group_bv_select_batch(self._h, self._handles(replicas), bit, _ptr(i), n, _ptr(out), chunks))
        return out

    def csa_from_text(self, text, flags: int = 0):
        """One csa_wt per device from one text (host memory or devices[0]): one broadcast, every device lays out its own."""
        t = _bytes_arg(text, "text")
        n = len(self.devices)
        reps = (C.c_void_p * n)()
        capi.check(capi.lib().sdsl_hip_group_fm_create_from_text(self._h, _ptr(t), t.numel() if _is_tensor(t) else t.size, flags, reps))
        out = []
        for r in range(n):
            o = csa_wt.__new__(csa_wt)
            _Handle.__init__(o)
            o._h = C.c_void_p(reps[r])
            o.device = self.devices[r]
            out.append(o)
        return out

    def count(self, replicas, patterns, m: int, out=None, chunks: int = 4):
        p = _bytes_arg(patterns, "patterns")
        n = (p.numel() if _is_tensor(p) else p.size) // m
        out = _out_for(p, n, np.uint64, out)
        capi.check(capi.lib().sdsl_hip_group_fm_count_batch(self._h, self._handles(replicas), _ptr(p), m, n, _ptr(out), chunks))
        return out
