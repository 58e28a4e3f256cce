"""Open-ended version of test_gpu_fuzz.py: python tests/gpu_fuzz_long.py [seconds] [seed0]
Draws fresh seeds until the time budget is spent; every case is checked against the oracle."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import lz77_amd as L
import oracle_lib as O
from test_gpu_fuzz import _cases, _make

import random
os.environ.setdefault("LZ77X_FAKE_DEVICES", "4")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t_end = time.time() + budget
runs = 0
rng = random.Random(seed)
while time.time() < t_end:
    for sb, la, n, alpha, mode, s in _cases(seed, 20):
        if rng.random() < 0.15 and sb <= 8193:
            n = rng.randint(100_000, 500_000)                 # several blocks of the recurrence, several tiles, long runs
        data = _make(n, alpha, mode, s)
        want = O.encode_bst(data, sb, la)
        # the encoder in one segment or a few, the decoder in one range or many (the knobs never change a byte)
        for k in ("LZ77X_SEGMENT", "LZ77X_DECODE_RANGE", "LZ77X_DECODE_RANGE_BYTES", "LZ77X_SHARD_STRETCH", "LZ77X_PRIO_MAX_ITERS", "LZ77X_TOKEN_CHUNK"):
            os.environ.pop(k, None)
        # round 6: several token launches per segment now and then (the large windows build their hand-overs by rank per launch:
        # other regions, other offsets)
        if rng.random() < 0.2:
            os.environ["LZ77X_TOKEN_CHUNK"] = str(rng.choice([20000, 90000, 300000]))
        if rng.random() < 0.3:
            os.environ["LZ77X_SEGMENT"] = str(rng.choice([1, 30000, 100000]))
        # round 5: one stream over 2-4 contexts in stretches, and the recurrence of a segment / a stretch on a host core
        # (the path an error front takes) now and then
        shards = rng.choice([1, 1, 1, 2, 3, 4])
        if shards > 1:
            os.environ["LZ77X_SHARD_STRETCH"] = str(rng.choice([20000, 70000, 10 ** 9]))
        if rng.random() < 0.15:
            os.environ["LZ77X_PRIO_MAX_ITERS"] = str(rng.choice([1, 2, 7]))
        L.lib().lz77x_set_shards(shards)
        try:
            got = L.encode(data, la, sb)
        finally:
            L.lib().lz77x_set_shards(1)
        assert got == want, (seed, sb, la, n, alpha, mode, s, shards, {k: v for k, v in os.environ.items() if k.startswith("LZ77X_")})
        r = rng.random()
        if r < 0.3:
            os.environ["LZ77X_DECODE_RANGE"] = str(rng.choice([8, 64, 1000, 20000]))
        elif r < 0.45:
            os.environ["LZ77X_DECODE_RANGE_BYTES"] = str(rng.choice([3000, 40000]))
        # round 5: the decoder over 2-4 contexts too, in stretches of tokens now and then
        dshards = rng.choice([1, 1, 2, 3, 4])
        os.environ.pop("LZ77X_DECODE_SHARD_STRETCH", None)
        if dshards > 1 and rng.random() < 0.5:
            os.environ["LZ77X_DECODE_SHARD_STRETCH"] = str(rng.choice([4000, 30000]))
        L.lib().lz77x_set_shards(dshards)
        try:
            dec = L.decode(want)
        finally:
            L.lib().lz77x_set_shards(1)
        assert dec == (data.tobytes() if sb & (sb - 1) else O.decode(want)), (seed, sb, la, n, alpha, mode, s, os.environ.get("LZ77X_DECODE_RANGE"),
                                                                             os.environ.get("LZ77X_DECODE_RANGE_BYTES"), dshards, os.environ.get("LZ77X_DECODE_SHARD_STRETCH"))
        runs += 1
    seed += 1
print("fuzz ok: %d cases, seeds up to %d" % (runs, seed - 1))
