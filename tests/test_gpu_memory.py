"""Device memory: the encoder sizes its segments and the decoder its ranges to what the device has to spare
(device_budget in ctx.cpp); the output bytes never depend on it.  LZ77X_DEVICE_MEM_LIMIT caps what a call may plan
with, so a tight device can be imitated on the test box's 288 GB."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import lz77_amd as L
from lz77_amd import synth
from full_inputs import full_input

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = {r["name"]: r for r in json.load(open(os.path.join(HERE, "golden", "golden_full.json")))["full"]}


def _free():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _sha_dev(d_z, zn):
    h = hashlib.sha256()
    for at in range(0, zn, 1 << 28):
        h.update(d_z[at:min(at + (1 << 28), zn)].cpu().numpy().tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name,limit", [("S1", 2_000_000_000), ("S3", 12_000_000_000), ("S4", 9_000_000_000)])
def test_encode_and_decode_fit_a_memory_limit(name, limit, monkeypatch):
    """an encode that would hold 5.7 GB (S1), 30 GB (S3: s=65535 l=255) or 30 GB (S4: 1 GB) of device memory when left alone
    plans with a third of that: smaller match launches, smaller segments, two of them in flight -- the reference's digest,
    and the library's footprint (cached buffers, the peak of everything it allocated) stays below the limit.  The decode
    of the same stream fits the same limit."""
    import torch
    r = FULL[name]
    n, sb, la = r["n"], r["sb"], r["la"]
    data = full_input(r)
    d_in = torch.from_numpy(data).cuda()
    cap = L.encode_bound(n, la, sb)
    d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.lib().lz77x_shutdown()
    monkeypatch.setenv("LZ77X_DEVICE_MEM_LIMIT", str(limit))
    f0 = _free()
    zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
    stats = L.last_stats()
    held = f0 - _free()
    assert zn == r["zn"] and _sha_dev(d_z, zn) == r["sha256_lz"], "stream differs from the reference's"
    assert stats["host_stageb_ms"] == 0
    assert held <= limit, "encode held %.1f MB of device memory against a limit of %.1f MB" % (held / 1e6, limit / 1e6)
    L.lib().lz77x_shutdown()
    monkeypatch.setenv("LZ77X_DEVICE_MEM_LIMIT", str(limit // 8))
    f0 = _free()
    assert L.decode_device(d_z.data_ptr(), zn, d_back.data_ptr(), n, st) == n
    held = f0 - _free()
    assert bool(torch.equal(d_back, d_in))
    assert L.last_stats()["match_launches"] >= 2, "the limit did not cut the stream into ranges"
    assert held <= limit // 8, "decode held %.1f MB against %.1f MB" % (held / 1e6, limit / 8e6)
    L.lib().lz77x_shutdown()


def test_default_footprint_is_bounded():
    """what the library holds after an S4 encode (1 GB, one segment) and after the decode of its stream, left alone:
    27-31 bytes per input byte for the encode (tools/mem_probe.py), a few bytes per output byte for the decode"""
    import torch
    r = FULL["S4"]
    n, sb, la = r["n"], r["sb"], r["la"]
    d_in = torch.from_numpy(full_input(r)).cuda()
    cap = L.encode_bound(n, la, sb)
    d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.lib().lz77x_shutdown()
    f0 = _free()
    zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
    held_enc = f0 - _free()
    assert zn == r["zn"]
    assert held_enc <= 34 * n, "encode holds %.1f GB" % (held_enc / 1e9)
    L.lib().lz77x_shutdown()
    f0 = _free()
    assert L.decode_device(d_z.data_ptr(), zn, d_in.data_ptr(), n, st) == n
    held_dec = f0 - _free()
    assert held_dec <= 4 * n, "decode holds %.1f GB" % (held_dec / 1e9)
    L.lib().lz77x_shutdown()


def test_four_files_of_1gb_at_once(tmp_path):
    """lz77x_encode_files with four 1 GB files, each on a context set of its own: four segments of 10^9 positions
    resident together (~30 GB each); every stream carries the reference's digest, the footprint stays below 4 x 34 GB"""
    import torch                                            # noqa: F401
    r = FULL["S4"]
    n, sb, la = r["n"], r["sb"], r["la"]
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    d = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    src = os.path.join(d, "lz77x_four.in")
    outs = [os.path.join(d, "lz77x_four_%d.lz" % i) for i in range(4)]
    try:
        full_input(r).tofile(src)
        L.lib().lz77x_shutdown()
        f0 = _free()
        FP = ctypes.c_void_p * 4
        fi = FP(*[libc.fopen(src.encode(), b"rb") for _ in range(4)])
        fo = FP(*[libc.fopen(p.encode(), b"wb") for p in outs])
        rcs = (ctypes.c_int * 4)()
        rc = L.lib().lz77x_encode_files(4, fi, fo, la, sb, rcs)
        for f in list(fi) + list(fo):
            libc.fclose(f)
        held = f0 - _free()
        assert rc == 0 and list(rcs) == [0] * 4, L.lib().lz77x_last_error()
        assert held <= 4 * 34 * n, "four contexts hold %.1f GB" % (held / 1e9)
        for p in outs:
            h = hashlib.sha256()
            with open(p, "rb") as f:
                while True:
                    b = f.read(1 << 26)
                    if not b:
                        break
                    h.update(b)
            assert h.hexdigest() == r["sha256_lz"], p
    finally:
        L.lib().lz77x_shutdown()
        for p in [src] + outs:
            if os.path.exists(p):
                os.remove(p)


def test_out_of_device_memory_fails_cleanly():
    """a device with (almost) nothing to spare: the call returns LZ77X_E_HIP with a message instead of faulting, and the
    library works again once memory is back"""
    import torch
    data = synth.text(50_000_000, 5)
    L.lib().lz77x_shutdown()
    free = _free()
    hog = torch.empty(free - (150 << 20), dtype=torch.uint8, device="cuda")     # leave ~150 MB
    try:
        with pytest.raises(L.Lz77Error) as e:
            L.encode(data)
        assert e.value.code == -3, e.value
    finally:
        del hog
        torch.cuda.empty_cache()
    L.lib().lz77x_shutdown()
    z = L.encode(data[:3_000_000])
    assert L.decode(z) == data[:3_000_000].tobytes()
