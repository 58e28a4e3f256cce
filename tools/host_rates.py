"""PCIe-inclusive rates of the buffer-level entry points (host memory in, host memory out): python tools/host_rates.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lz77_amd as L
from lz77_amd import synth
out = []
for kind, n in (("text", 100_000_000), ("text", 1_000_000_000)):
    data = synth.make(kind, n, synth.SEED_S1 if n < 1e9 else synth.SEED_S4)
    best_e = best_d = 1e9
    for it in range(3):
        t0 = time.perf_counter(); z = L.encode(data); t1 = time.perf_counter()
        back = L.decode(z); t2 = time.perf_counter()
        if it:
            best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    assert back == data.tobytes()
    # the C calls alone (no copy into a Python object)
    import ctypes
    lib = L.lib()
    a = data
    raw_e = raw_d = 1e9
    for it in range(3):
        zout = ctypes.POINTER(ctypes.c_uint8)(); zn = ctypes.c_size_t(0)
        t0 = time.perf_counter(); rc = lib.lz77x_encode(a.ctypes.data, a.size, -1, -1, ctypes.byref(zout), ctypes.byref(zn)); t1 = time.perf_counter()
        assert rc == 0
        o2 = ctypes.POINTER(ctypes.c_uint8)(); n2 = ctypes.c_size_t(0)
        t2 = time.perf_counter(); rc = lib.lz77x_decode(zout, zn.value, ctypes.byref(o2), ctypes.byref(n2)); t3 = time.perf_counter()
        assert rc == 0 and n2.value == n
        lib.lz77x_free(zout); lib.lz77x_free(o2)
        if it:
            raw_e, raw_d = min(raw_e, t1 - t0), min(raw_d, t3 - t2)
    rec = {"kind": kind, "bytes": n, "encode_ms": round(best_e * 1e3, 1), "decode_ms": round(best_d * 1e3, 1),
           "encode_GBps": round(n / best_e / 1e9, 2), "decode_GBps": round(n / best_d / 1e9, 2),
           "c_call_encode_ms": round(raw_e * 1e3, 1), "c_call_decode_ms": round(raw_d * 1e3, 1),
           "c_call_encode_GBps": round(n / raw_e / 1e9, 2), "c_call_decode_GBps": round(n / raw_d / 1e9, 2),
           "note": "encode_ms / decode_ms: lz77_amd.encode / decode (the result is also copied into a Python bytes object); c_call_*: lz77x_encode / lz77x_decode alone"}
    print(rec, flush=True)
    out.append(rec)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "host_rates.json"), "w"), indent=1)
