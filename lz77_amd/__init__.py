"""lz77_amd -- Python host-side mirror of the C ABI in include/lz77_mi355x.h.

The product is liblz77_mi355x.so (hand-written HIP for gfx950 + a C host layer) and the
`lz77` CLI next to it; this module only binds that library with ctypes so tests and
bench.py can drive it.  The function names follow the reference's interface
(lz77.h:14-15): encode(data, la, sb) / decode(stream), la BEFORE sb, -1 = default.

There is no CPU fallback: if the library or a GPU is missing every call raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

__all__ = ["encode", "decode", "encode_c", "decode_c", "CBuffer", "encode_device", "decode_device", "encode_path", "decode_path", "last_stats", "build", "lib",
           "Lz77Error", "LIB_PATH", "CLI_PATH"]

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# LZ77X_TEST_LIB / LZ77X_TEST_CLI: tests/test_sanitize_cpu.py points the binding at the -fsanitize=address,undefined build
# of the same sources (csrc/Makefile `asan`); nothing else sets them
LIB_PATH = os.environ.get("LZ77X_TEST_LIB") or os.path.join(_HERE, "liblz77_mi355x.so")
CLI_PATH = os.environ.get("LZ77X_TEST_CLI") or os.path.join(_HERE, "lz77")

DEFAULT_LA = 15      # lz77.c:21
DEFAULT_SB = 4095    # lz77.c:22

E_CAP = -6


class Lz77Error(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        super().__init__("%s (code %d)%s" % (what, code, (": " + detail) if detail else ""))


class Stats(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double), ("k_match_ms", ctypes.c_double), ("k_sort_ms", ctypes.c_double),
                ("k_token_ms", ctypes.c_double),
                ("k_decode_ms", ctypes.c_double), ("host_chain_ms", ctypes.c_double),
                ("host_stageb_ms", ctypes.c_double), ("copy_ms", ctypes.c_double),
                ("n", ctypes.c_uint64), ("zn", ctypes.c_uint64), ("ntok", ctypes.c_uint64),
                ("transfers", ctypes.c_uint64), ("match_launches", ctypes.c_uint32),
                ("decode_rounds", ctypes.c_uint32), ("k_walk_ms", ctypes.c_double), ("k_tiebreak_ms", ctypes.c_double),
                ("token_launches", ctypes.c_uint32), ("prio_iters", ctypes.c_uint32),
                ("k_prio_ms", ctypes.c_double), ("k_chain_ms", ctypes.c_double), ("k_prio_fwd_ms", ctypes.c_double),
                ("k_prio_back_ms", ctypes.c_double), ("k_prio_scan_ms", ctypes.c_double), ("k_sort_chunks_ms", ctypes.c_double)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950); works without a GPU."""
    srcdir = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", srcdir, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", srcdir])
    return LIB_PATH



_sz = ctypes.c_size_t
_vp = ctypes.c_void_p
_u8pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))

# every symbol include/lz77_mi355x.h declares, with its ctypes signature
SYMBOLS = {
    "lz77x_encode": (ctypes.c_int, [_vp, _sz, ctypes.c_int, ctypes.c_int, _u8pp, ctypes.POINTER(_sz)]),
    "lz77x_decode": (ctypes.c_int, [_vp, _sz, _u8pp, ctypes.POINTER(_sz)]),
    "lz77x_free": (None, [_vp]),
    "lz77x_encode_bound": (_sz, [_sz, ctypes.c_int, ctypes.c_int]),
    "lz77x_encode_device": (ctypes.c_int, [_vp, _sz, ctypes.c_int, ctypes.c_int, _vp, _sz, ctypes.POINTER(_sz), _vp]),
    "lz77x_decode_device": (ctypes.c_int, [_vp, _sz, _vp, _sz, ctypes.POINTER(_sz), _vp]),
    "lz77x_encode_file": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "lz77x_decode_file": (ctypes.c_int, [_vp, _vp]),
    "lz77x_set_shards": (ctypes.c_int, [ctypes.c_int]),
    "lz77x_device_count": (ctypes.c_int, []),
    "lz77x_shutdown": (None, []),
    "lz77x_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "lz77x_last_error": (ctypes.c_char_p, []),
    "lz77x_version": (ctypes.c_char_p, []),
    "lz77x_last_stats": (ctypes.c_int, [ctypes.POINTER(Stats)]),
    "lz77x_stage_maxlen": (ctypes.c_int, [_vp, _sz, ctypes.c_int, ctypes.c_int, _vp]),
    "lz77x_stage_neighbours": (ctypes.c_int, [_vp, _sz, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "lz77x_stage_priorities": (ctypes.c_int, [_vp, _vp, _sz, ctypes.c_int, _vp]),
    "lz77x_stage_priorities_device": (ctypes.c_int, [_vp, _vp, _sz, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_int)]),
    "lz77x_stage_chain_device": (ctypes.c_int, [_vp, _sz, ctypes.c_int, _vp, ctypes.POINTER(_sz)]),
    "lz77x_shard_plan": (ctypes.c_int, [_sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "lz77x_shard_compose_cells": (None, [_vp, _vp, ctypes.c_int, _vp]),
    "lz77x_shard_compose_chain": (None, [_vp, _vp, _vp, _vp]),
    "lz77x_shard_token_cut": (ctypes.c_uint64, [ctypes.c_uint64, ctypes.c_int, ctypes.c_int]),
    "lz77x_shard_compose_tail": (None, [_vp, ctypes.c_int, _vp, _vp]),
    "lz77x_shard_compose_tail32": (None, [_vp, ctypes.c_int, _vp, _vp]),
    "lz77x_encode_files": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "lz77x_decode_files": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp]),
}


VARIANTS_LIB_PATH = os.path.join(_HERE, "liblz77_mi355x_variants.so")
_libs = {}
_variants = False
_last_path = None


def use_variants(on: bool) -> None:
    """Tests only: route every call of this module through liblz77_mi355x_variants.so -- the build with the cross-check
    kernels of earlier rounds, the timing probes and the knobs that force fallbacks (LZ77X_MATCH_VARIANT, ...).  The
    product library does not contain them and does not read those knobs."""
    global _variants
    _variants = bool(on)


# the knobs only the variants build reads (csrc/lz77x_internal.h, LZ77X_VENV)
VARIANT_KNOBS = ("LZ77X_MATCH_VARIANT", "LZ77X_TOKEN_VARIANT", "LZ77X_SORT_VARIANT", "LZ77X_DECODE_V1", "LZ77X_DECODE_VARIANT",
                 "LZ77X_XFER_V1", "LZ77X_WALK_BIG_V1", "LZ77X_TOKENS_BUCKET", "LZ77X_C1_SORT_V1", "LZ77X_BIG_SORT_V1",
                 "LZ77X_PRIO_BACK_SWEEP", "LZ77X_PW_PREP_V1", "LZ77X_PW_PROBE", "LZ77X_PW_DEBUG", "LZ77X_WALK_DEBUG", "LZ77X_SERIAL", "LZ77X_SPLIT",
                 "LZ77X_CHAIN_STREAM", "LZ77X_PRIO_SORTCAP", "LZ77X_PRIO_WIDE", "LZ77X_HOST_STAGEB", "LZ77X_TS_ENTCAP", "LZ77X_TS_V4", "LZ77X_TS_PROBE", "LZ77X_TS_BIG", "LZ77X_DECODE_UNFUSED", "LZ77X_TS_OVERLAP_PROBE", "LZ77X_WALK_FRINGE_V4", "LZ77X_RANK_PROBE", "LZ77X_NO_SHORT_INDEX", "LZ77X_RANK_LPT", "LZ77X_NO_RANK_INDEX")


def lib():
    """dlopen liblz77_mi355x.so (fails loudly if it has not been built).  A process that sets one of VARIANT_KNOBS -- a
    cross-check test -- gets the variants build for that call; nothing else does."""
    global _last_path
    path = VARIANTS_LIB_PATH if (_variants or any(k in os.environ for k in VARIANT_KNOBS)) else LIB_PATH
    if _last_path is not None and _last_path != path and _last_path in _libs:
        # a test process that goes back and forth between the two builds: the one it leaves gives its cached device
        # contexts and buffers back (each library caches its own -- gigabytes after a large encode)
        _libs[_last_path].lz77x_shutdown()
    _last_path = path
    L = _libs.get(path)
    if L is None:
        if not os.path.exists(path):
            raise Lz77Error(-4, os.path.basename(path) + " is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        # torch ships its own libamdhip64.so.7; whichever HIP runtime is mapped first serves the
        # whole process, and device pointers are only shareable within ONE runtime.  Let torch
        # (the memory/stream plumbing of tests and bench.py) map its copy before we bind ours.
        if os.environ.get("LZ77X_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _libs[path] = L
    return L


def _check(rc: int):
    if rc != 0:
        L = lib()
        raise Lz77Error(rc, L.lz77x_strerror(rc).decode(), L.lz77x_last_error().decode())


def _u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def encode(data, la: int = -1, sb: int = -1) -> bytes:
    """encode(file, out, la, sb) of lz77.c:51 on an in-memory buffer -> compressed stream."""
    a = _u8(data)
    out = ctypes.POINTER(ctypes.c_uint8)()
    zn = _sz(0)
    _check(lib().lz77x_encode(a.ctypes.data, a.size, int(sb), int(la), ctypes.byref(out), ctypes.byref(zn)))
    try:
        return ctypes.string_at(out, zn.value)
    finally:
        lib().lz77x_free(out)


class CBuffer:
    """The malloc'ed result of lz77x_encode / lz77x_decode as it left the C ABI: `.view` is a numpy array over the library's own
    buffer (no copy into a Python bytes object), released through lz77x_free by close() / the context manager / the finalizer."""

    def __init__(self, lib_, ptr, n):
        self._lib, self._ptr, self.n = lib_, ptr, int(n)
        self.view = np.ctypeslib.as_array(ptr, shape=(self.n,)) if self.n else np.zeros(0, dtype=np.uint8)

    def close(self):
        if self._ptr is not None:
            self.view = None
            self._lib.lz77x_free(self._ptr)
            self._ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def tobytes(self) -> bytes:
        return self.view.tobytes()


def encode_c(data, la: int = -1, sb: int = -1) -> CBuffer:
    """lz77x_encode exactly as a C caller sees it: the stream stays in the buffer the library allocated (bench.py times this,
    not encode(), whose copy into a bytes object costs a 1 GB stream ~200 ms of Python)."""
    a = _u8(data)
    out = ctypes.POINTER(ctypes.c_uint8)()
    zn = _sz(0)
    L = lib()
    _check(L.lz77x_encode(a.ctypes.data, a.size, int(sb), int(la), ctypes.byref(out), ctypes.byref(zn)))
    return CBuffer(L, out, zn.value)


def decode_c(stream) -> CBuffer:
    """lz77x_decode as a C caller sees it (see encode_c); `stream` may be a CBuffer's view."""
    a = _u8(stream)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = _sz(0)
    L = lib()
    _check(L.lz77x_decode(a.ctypes.data, a.size, ctypes.byref(out), ctypes.byref(n)))
    return CBuffer(L, out, n.value)


def decode(stream) -> bytes:
    """decode(file, out) of lz77.c:148 on an in-memory stream -> original bytes."""
    a = _u8(stream)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = _sz(0)
    _check(lib().lz77x_decode(a.ctypes.data, a.size, ctypes.byref(out), ctypes.byref(n)))
    try:
        return ctypes.string_at(out, n.value)
    finally:
        lib().lz77x_free(out)


_libc = None


def _fopen(path: str, mode: bytes):
    """a FILE* of the C library for the FILE* entry points (lz77.h:14-15 take FILE*, main.c:141-158 opens them)"""
    global _libc
    if _libc is None:
        _libc = ctypes.CDLL(None)
        _libc.fopen.restype = ctypes.c_void_p
        _libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        _libc.fclose.argtypes = [ctypes.c_void_p]
    f = _libc.fopen(os.fsencode(path), mode)
    if not f:
        raise OSError("cannot open " + path)
    return f


def encode_path(src: str, dst: str, la: int = -1, sb: int = -1) -> None:
    """lz77x_encode_file on two paths: what `lz77 -c -i src -o dst` does inside this process"""
    fi, fo = _fopen(src, b"rb"), _fopen(dst, b"wb")
    try:
        rc = lib().lz77x_encode_file(fi, fo, int(la), int(sb))
    finally:
        _libc.fclose(fi)
        _libc.fclose(fo)
    _check(rc)


def decode_path(src: str, dst: str) -> None:
    """lz77x_decode_file on two paths: what `lz77 -d -i src -o dst` does inside this process"""
    fi, fo = _fopen(src, b"rb"), _fopen(dst, b"wb")
    try:
        rc = lib().lz77x_decode_file(fi, fo)
    finally:
        _libc.fclose(fi)
        _libc.fclose(fo)
    _check(rc)


def encode_bound(n: int, la: int = -1, sb: int = -1) -> int:
    return int(lib().lz77x_encode_bound(n, int(sb), int(la)))


def encode_device(d_in: int, n: int, d_out: int, out_cap: int, la: int = -1, sb: int = -1, stream: int = 0) -> int:
    """Device-resident encode: raw device pointers (e.g. torch.Tensor.data_ptr()); returns stream size."""
    zn = _sz(0)
    _check(lib().lz77x_encode_device(d_in, n, int(sb), int(la), d_out, out_cap, ctypes.byref(zn), stream))
    return zn.value


def decoded_size_device(d_z: int, zn: int, stream: int = 0) -> int:
    n = _sz(0)
    _check(lib().lz77x_decode_device(d_z, zn, None, 0, ctypes.byref(n), stream))
    return n.value


def decode_device(d_z: int, zn: int, d_out: int, out_cap: int, stream: int = 0) -> int:
    n = _sz(0)
    _check(lib().lz77x_decode_device(d_z, zn, d_out, out_cap, ctypes.byref(n), stream))
    return n.value


def last_stats() -> dict:
    st = Stats()
    _check(lib().lz77x_last_stats(ctypes.byref(st)))
    return st.as_dict()


def stage_maxlen(data, la: int = -1, sb: int = -1) -> np.ndarray:
    a = _u8(data)
    la = DEFAULT_LA if la == -1 else la
    sb = DEFAULT_SB if sb == -1 else sb
    out = np.zeros(max(a.size, 1), dtype=np.uint8)
    _check(lib().lz77x_stage_maxlen(a.ctypes.data, a.size, sb, la, out.ctypes.data))
    return out[:a.size]


def stage_neighbours(data, la: int = -1, sb: int = -1):
    a = _u8(data)
    la = DEFAULT_LA if la == -1 else la
    sb = DEFAULT_SB if sb == -1 else sb
    P = np.zeros(max(a.size, 1), dtype=np.uint16)
    S = np.zeros(max(a.size, 1), dtype=np.uint16)
    _check(lib().lz77x_stage_neighbours(a.ctypes.data, a.size, sb, la, P.ctypes.data, S.ctypes.data))
    return P[:a.size], S[:a.size]


def stage_priorities(P: np.ndarray, S: np.ndarray, sb: int) -> np.ndarray:
    P = np.ascontiguousarray(P, dtype=np.uint16)
    S = np.ascontiguousarray(S, dtype=np.uint16)
    xval = np.empty(max(P.size, 1), dtype=np.uint32)
    _check(lib().lz77x_stage_priorities(P.ctypes.data, S.ctypes.data, P.size, int(sb), xval.ctypes.data))
    return xval[:P.size]


def stage_priorities_device(P: np.ndarray, S: np.ndarray, sb: int):
    """-> (xval, iterations) from the device form of the recurrence (k_prio.hip)"""
    P = np.ascontiguousarray(P, dtype=np.uint16)
    S = np.ascontiguousarray(S, dtype=np.uint16)
    xval = np.empty(max(P.size, 1), dtype=np.uint32)
    it = ctypes.c_int(0)
    _check(lib().lz77x_stage_priorities_device(P.ctypes.data, S.ctypes.data, P.size, int(sb), xval.ctypes.data, ctypes.byref(it)))
    return xval[:P.size], it.value


def stage_chain_device(maxlen: np.ndarray, la: int) -> np.ndarray:
    """-> chain positions from the device parse chain (k_chain.hip)"""
    m = np.ascontiguousarray(maxlen, dtype=np.uint8)
    chain = np.empty(max(m.size, 1), dtype=np.uint32)
    nt = _sz(0)
    _check(lib().lz77x_stage_chain_device(m.ctypes.data, m.size, int(la), chain.ctypes.data, ctypes.byref(nt)))
    return chain[:nt.value]
