"""Soak test of the encode pipeline's flow control: random inputs x random chunk / group / ring /
shard settings, each checked against the oracle.  python tests/gpu_stress.py [seconds] [seed]"""
import os, sys, time, random
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import lz77_amd as L
import oracle_lib as O
from lz77_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
os.environ["LZ77X_FAKE_DEVICES"] = "4"
t_end = time.time() + budget
runs = 0
while time.time() < t_end:
    kind = rng.choice(["text", "random", "mixed", "lowent", "code"])
    n = rng.choice([0, 1, 100, 70_000, 300_000, 1_000_000, 2_500_000, 6_000_000])
    sb, la = rng.choice([(4095, 15), (4095, 15), (1000, 10), (255, 7), (65535, 255), (8192, 16), (2048, 31)])
    if sb > 8192:
        n = min(n, 1_200_000)
    env = {"LZ77X_CHUNK_REGIONS": str(rng.choice([1, 2, 3, 8, 64, 512])), "LZ77X_MATCH_GROUP": str(rng.choice([1, 2, 3, 8])),
           "LZ77X_RING_SLOTS": str(rng.choice([3, 4, 5, 8, 32]))}
    shards = rng.choice([1, 1, 2, 3, 4])
    os.environ.update(env)
    L.lib().lz77x_set_shards(shards)
    data = synth.make(kind, n, rng.randrange(1 << 30))
    want = O.encode_bst(data, sb, la)
    got = L.encode(data, la, sb)
    assert got == want, (kind, n, sb, la, env, shards)
    if sb & (sb - 1):
        assert L.decode(got) == data.tobytes()
    runs += 1
print("stress ok: %d runs" % runs)
