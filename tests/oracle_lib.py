"""ctypes binding of oracle/liblz77_oracle.so (TEST INFRASTRUCTURE -- see oracle/lz77_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liblz77_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "lz77_ref")
BAD = (1 << 64) - 1
NONE32 = 0xFFFFFFFF

_sz = ctypes.c_size_t
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "lz77_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liblz77_oracle.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.lz77o_bitof.restype = ctypes.c_int
        for name in ("lz77o_bound", "lz77o_encode_bst", "lz77o_encode_model", "lz77o_decode",
                     "lz77o_tokens", "lz77o_stage_b"):
            getattr(L, name).restype = _sz
        _lib = L
    return _lib


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def _ptr(a: np.ndarray, ty=ctypes.c_uint8):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def bitof(n: int) -> int:
    return lib().lz77o_bitof(int(n))


def token_bits(sb: int, la: int) -> int:
    return bitof(sb) + bitof(la) + 8


def _encode(fn, data, sb, la) -> bytes:
    a = _as_u8(data)
    cap = int(lib().lz77o_bound(_sz(a.size), sb, la)) + 8
    out = np.empty(cap, dtype=np.uint8)
    r = fn(_ptr(a), _sz(a.size), int(sb), int(la), _ptr(out), _sz(cap))
    if r == BAD:
        raise ValueError("oracle encode failed (bad arguments?)")
    return out[:r].tobytes()


def encode_bst(data, sb=4095, la=15) -> bytes:
    return _encode(lib().lz77o_encode_bst, data, sb, la)


def encode_model(data, sb=4095, la=15) -> bytes:
    return _encode(lib().lz77o_encode_model, data, sb, la)


def decode(z) -> bytes:
    a = _as_u8(z)
    n = lib().lz77o_decode(_ptr(a), _sz(a.size), None, _sz(0))
    if n == BAD:
        raise ValueError("oracle decode failed")
    out = np.empty(max(int(n), 1), dtype=np.uint8)
    r = lib().lz77o_decode(_ptr(a), _sz(a.size), _ptr(out), _sz(n))
    assert r == n
    return out[:n].tobytes()


def tokens(z):
    """-> (sb, la, off[int32], len[int32], next[uint8])"""
    a = _as_u8(z)
    sb = ctypes.c_int(0)
    la = ctypes.c_int(0)
    ntok = lib().lz77o_tokens(_ptr(a), _sz(a.size), ctypes.byref(sb), ctypes.byref(la), None, None, None, _sz(0))
    if ntok == BAD:
        raise ValueError("bad stream")
    off = np.empty(ntok, dtype=np.int32)
    ln = np.empty(ntok, dtype=np.int32)
    nx = np.empty(ntok, dtype=np.uint8)
    lib().lz77o_tokens(_ptr(a), _sz(a.size), ctypes.byref(sb), ctypes.byref(la),
                       _ptr(off, ctypes.c_int32), _ptr(ln, ctypes.c_int32), _ptr(nx), _sz(ntok))
    return sb.value, la.value, off, ln, nx


def maxlen(data, sb=4095, la=15) -> np.ndarray:
    a = _as_u8(data)
    out = np.zeros(max(a.size, 1), dtype=np.uint8)
    lib().lz77o_maxlen(_ptr(a), _sz(a.size), int(sb), int(la), _ptr(out))
    return out[:a.size]


def stage_a(data, sb=4095, la=15, tree: bool = False):
    """-> (P, S[, two]) uint16 distances, 0 = none"""
    a = _as_u8(data)
    P = np.zeros(max(a.size, 1), dtype=np.uint16)
    S = np.zeros(max(a.size, 1), dtype=np.uint16)
    if tree:
        two = np.zeros(max(a.size, 1), dtype=np.uint8)
        lib().lz77o_stage_a_tree(_ptr(a), _sz(a.size), int(sb), int(la),
                                 _ptr(P, ctypes.c_uint16), _ptr(S, ctypes.c_uint16), _ptr(two))
        return P[:a.size], S[:a.size], two[:a.size]
    lib().lz77o_stage_a(_ptr(a), _sz(a.size), int(sb), int(la),
                        _ptr(P, ctypes.c_uint16), _ptr(S, ctypes.c_uint16))
    return P[:a.size], S[:a.size]


def stage_b(P: np.ndarray, S: np.ndarray, sb: int) -> np.ndarray:
    n = P.size
    P = np.ascontiguousarray(P, dtype=np.uint16)
    S = np.ascontiguousarray(S, dtype=np.uint16)
    xval = np.empty(max(n, 1), dtype=np.uint32)
    r = lib().lz77o_stage_b(_ptr(P, ctypes.c_uint16), _ptr(S, ctypes.c_uint16), _sz(n), int(sb),
                            _ptr(xval, ctypes.c_uint32))
    assert r != BAD
    return xval[:n]


def splitmix_fill(seed: int, n: int) -> np.ndarray:
    out = np.empty(max(n, 1), dtype=np.uint8)
    lib().lz77o_splitmix_fill(ctypes.c_uint64(seed), _ptr(out), _sz(n))
    return out[:n]


# ---- the compiled reference (only exists where /root/reference was available at build time) ----

def have_ref() -> bool:
    return os.path.exists(REF_BIN)


def ref_encode(data, sb=4095, la=15, tmpdir="/tmp") -> bytes:
    a = _as_u8(data)
    fin = os.path.join(tmpdir, "lz77ref_%d.in" % os.getpid())
    fout = os.path.join(tmpdir, "lz77ref_%d.lz" % os.getpid())
    a.tofile(fin)
    try:
        subprocess.check_call([REF_BIN, "-c", "-i", fin, "-o", fout, "-s", str(sb), "-l", str(la)])
        with open(fout, "rb") as f:
            return f.read()
    finally:
        for p in (fin, fout):
            if os.path.exists(p):
                os.unlink(p)


def ref_decode(z, tmpdir="/tmp") -> bytes:
    fin = os.path.join(tmpdir, "lz77ref_%d.lz" % os.getpid())
    fout = os.path.join(tmpdir, "lz77ref_%d.out" % os.getpid())
    _as_u8(z).tofile(fin)
    try:
        subprocess.check_call([REF_BIN, "-d", "-i", fin, "-o", fout])
        with open(fout, "rb") as f:
            return f.read()
    finally:
        for p in (fin, fout):
            if os.path.exists(p):
                os.unlink(p)
