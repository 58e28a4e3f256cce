"""Open-ended version of test_gpu_fuzz.py: python tests/gpu_fuzz_long.py [seconds] [seed0]
Draws fresh seeds until the time budget is spent; every case is checked against the oracle."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import lz77_amd as L
import oracle_lib as O
from test_gpu_fuzz import _cases, _make

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t_end = time.time() + budget
runs = 0
while time.time() < t_end:
    for sb, la, n, alpha, mode, s in _cases(seed, 20):
        data = _make(n, alpha, mode, s)
        want = O.encode_bst(data, sb, la)
        got = L.encode(data, la, sb)
        assert got == want, (seed, sb, la, n, alpha, mode, s)
        dec = L.decode(want)
        if sb & (sb - 1):
            assert dec == data.tobytes(), (seed, sb, la, n, alpha, mode, s)
        else:
            assert len(dec) == data.size, (seed, sb, la, n, alpha, mode, s)
        runs += 1
    seed += 1
print("fuzz ok: %d cases, seeds up to %d" % (runs, seed - 1))
