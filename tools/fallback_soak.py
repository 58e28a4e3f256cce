"""The hand-over from the device pipeline to the host-assisted one, many times in ONE process, between calls of
every other kind (other geometries, segments, shards over fake devices, decodes in ranges, both library builds), native stderr
visible: python tools/fallback_soak.py [seconds]      -- every stream against the first of its input (the first against the oracle)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lz77_amd as L
import oracle_lib as O
from lz77_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
t_end = time.time() + budget
rng = np.random.default_rng(606)
inputs = []
for period in (4096, 4095, 8190, 12292):
    r = np.random.default_rng(period)
    for n in (3_200_000, 1_700_000):
        inputs.append(np.tile(r.integers(0, 256, period, dtype=np.uint8), n // period + 1)[:n].copy())
want = [None] * len(inputs)
others = [(synth.make(k, n, 90 + i), sb, la) for i, (k, n, sb, la) in enumerate(
    [("text", 900_000, 4095, 15), ("mixed", 700_000, 65535, 255), ("lowent", 300_000, 1000, 10), ("code", 500_000, 8191, 15), ("random", 200_000, 255, 7)])]
zothers = [None] * len(others)
fallbacks = calls = 0
while time.time() < t_end:
    i = int(rng.integers(len(inputs)))
    z = L.encode(inputs[i])
    fallbacks += L.last_stats()["host_stageb_ms"] > 0
    h = hashlib.sha256(z).hexdigest()
    if want[i] is None:
        want[i] = h
        assert z == O.encode_bst(inputs[i], 4095, 15), i
    assert h == want[i], (i, fallbacks)
    calls += 1
    j = int(rng.integers(len(others)))
    data, sb, la = others[j]
    env = {}
    m = int(rng.integers(5))
    if m == 1: env["LZ77X_SEGMENT"] = "250000"
    if m == 2: env.update(LZ77X_SHARDS="3", LZ77X_FAKE_DEVICES="3")
    if m == 3: env["LZ77X_DECODE_RANGE"] = "16000"
    if m == 4 and sb == 4095: env["LZ77X_TS_V4"] = "1"                  # (the variants build: the product library is shut down and re-created)
    for k, v in env.items(): os.environ[k] = v
    zz = L.encode(data, la, sb)
    if zothers[j] is None:
        zothers[j] = zz
        assert zz == O.encode_bst(data, sb, la), j
    assert zz == zothers[j], (j, env)
    assert L.decode(zz) == data.tobytes(), (j, env)
    for k in env: del os.environ[k]
    calls += 2
print("fallback soak ok: %d calls, %d hand-overs to the host-assisted pipeline" % (calls, fallbacks))
