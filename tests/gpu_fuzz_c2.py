"""Open-ended fuzz of the LARGE windows (sb > 4096: k_big_*, k_walk_wave, k_pw_*, k_tokens_rank_group with the hand-overs by
rank of round 6): python tests/gpu_fuzz_c2.py [seconds] [seed0]
Inputs of several regions (0.2 - 1.5 MB), small alphabets and planted copies (long runs of equal candidates: the run cache, the
search of a run's ends, the oldest member by the window's ranks), several token chunks / segments / shards now and then; every
stream against the oracle's BST encoder (tree.c:62-243 restated).  tests/gpu_fuzz_long.py draws every geometry; this one only
the ones the reference is slowest at."""
import os, sys, time, random
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import lz77_amd as L
import oracle_lib as O
from test_gpu_fuzz import _make

os.environ.setdefault("LZ77X_FAKE_DEVICES", "4")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
t_end = time.time() + budget
rng = random.Random(seed)
runs = 0
KNOBS = ("LZ77X_SEGMENT", "LZ77X_TOKEN_CHUNK", "LZ77X_SHARD_STRETCH", "LZ77X_PRIO_MAX_ITERS")
while time.time() < t_end:
    sb = rng.choice([4097, 8192, 8193, 12000, 20000, 32768, 40000, 65535, 65535, 65535])
    la = rng.choice([2, 3, 9, 16, 40, 255, 255])
    n = rng.randint(200_000, 1_500_000)
    alpha = rng.choice([2, 3, 4, 16, 64, 256])                # (alpha 1 is a spine as deep as the window in the oracle's BST: minutes)
    mode = rng.choice(["iid", "runs", "copy", "copy"])
    s = rng.randrange(1 << 30)
    data = _make(n, alpha, mode, s)
    if rng.random() < 0.3:                                    # stretches of one byte between the data: runs of thousands of equal candidates
        at = 0
        r2 = np.random.default_rng(s)
        while at < n:
            ln = int(r2.integers(300, 1200))
            data[at:at + ln] = int(r2.integers(0, alpha))
            at += ln + int(r2.integers(2000, 60000))
    want = O.encode_bst(data, sb, la)
    for k in KNOBS:
        os.environ.pop(k, None)
    r = rng.random()
    if r < 0.25:
        os.environ["LZ77X_TOKEN_CHUNK"] = str(rng.choice([60000, 250000, 700000]))
    elif r < 0.45:
        os.environ["LZ77X_SEGMENT"] = str(rng.choice([300000, 600000]))
        os.environ["LZ77X_TOKEN_CHUNK"] = str(rng.choice([100000, 400000]))
    shards = rng.choice([1, 1, 1, 2, 3])
    if shards > 1 and n < shards * (4 * sb + 20000):
        shards = 1
    if shards > 1 and rng.random() < 0.5:
        os.environ["LZ77X_SHARD_STRETCH"] = str(rng.choice([400000, 10 ** 9]))
    if rng.random() < 0.1:
        os.environ["LZ77X_PRIO_MAX_ITERS"] = str(rng.choice([2, 7]))
    L.lib().lz77x_set_shards(shards)
    try:
        got = L.encode(data, la, sb)
    finally:
        L.lib().lz77x_set_shards(1)
    assert got == want, (seed, runs, sb, la, n, alpha, mode, s, shards, {k: v for k, v in os.environ.items() if k.startswith("LZ77X_")})
    assert L.decode(want) == (data.tobytes() if sb & (sb - 1) else O.decode(want)), (seed, runs, sb, la, n, alpha, mode, s)
    runs += 1
print("large-window fuzz ok: %d cases from seed %d" % (runs, seed))
