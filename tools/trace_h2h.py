"""lz77x_encode on 1 GB of text in host memory, three times, with the library's own trace (LZ77X_TRACE=1 in the environment): python tools/trace_h2h.py"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lz77_amd as L
from lz77_amd import synth
n = 1_000_000_000
data = synth.make("text", n, synth.SEED_S4)
lib = L.lib()
for it in range(3):
    if it == 2: os.environ["LZ77X_TRACE"] = "1"
    zout = ctypes.POINTER(ctypes.c_uint8)(); zn = ctypes.c_size_t(0)
    t0 = time.perf_counter(); rc = lib.lz77x_encode(data.ctypes.data, data.size, -1, -1, ctypes.byref(zout), ctypes.byref(zn)); t1 = time.perf_counter()
    print("encode %.1f ms" % ((t1 - t0) * 1e3), flush=True)
    lib.lz77x_free(zout)
