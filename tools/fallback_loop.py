"""Repeats the encode of the periodic inputs that take the host fallback and compares every stream with the first:
a loop to catch what a single run of tests/test_gpu_parity.py::test_periodic_input_reaches_the_fallback_without_a_knob misses.
python tools/fallback_loop.py [loops]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lz77_amd as L  # noqa: E402

loops = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 3_200_000
inputs = []
for period in (4096, 4095, 8190):
    rng = np.random.default_rng(period)
    inputs.append(np.tile(rng.integers(0, 256, period, dtype=np.uint8), n // period + 1)[:n].copy())
first = [None] * len(inputs)
fb = 0
for it in range(loops):
    for i, data in enumerate(inputs):
        z = L.encode(data)
        fb += L.last_stats()["host_stageb_ms"] > 0
        h = hashlib.sha256(z).hexdigest()
        if first[i] is None:
            first[i] = h
        assert h == first[i], (it, i)
    if it % 10 == 9:
        print("loop", it + 1, "ok", flush=True)
print("fallback loop ok: %d encodes, %d through the host path" % (loops * len(inputs), fb))
