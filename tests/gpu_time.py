"""Timing breakdown on the GPU box: python tests/gpu_time.py [kind] [n] [sb] [la]"""
import os, sys, time, hashlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import lz77_amd as L
from lz77_amd import synth

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
sb = int(sys.argv[3]) if len(sys.argv) > 3 else 4095
la = int(sys.argv[4]) if len(sys.argv) > 4 else 15
t = time.time(); data = synth.make(kind, n, 0x5EED0001); print("gen %.1fs" % (time.time() - t))
for it in range(int(os.environ.get('LZ77X_ITERS', '3'))):
    t = time.time(); z = L.encode(data, la, sb); te = time.time() - t
    se = L.last_stats()
    t = time.time(); back = L.decode(z); td = time.time() - t
    sd = L.last_stats()
    print("enc %.3fs (%.1f MB/s) dec %.3fs (%.1f MB/s) ratio %.4f" % (te, n / te / 1e6, td, n / td / 1e6, len(z) / n))
    print("  enc", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in se.items()})
    print("  dec", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in sd.items()})
assert back == data.tobytes()
print("roundtrip ok sha", hashlib.sha256(z).hexdigest()[:16])
# variants: sort-only probe and old token kernel
if os.environ.get('LZ77X_SWEEP', '1') != '1':
    sys.exit(0)
for name, env in (("match sort-only", {"LZ77X_MATCH_VARIANT": "2"}), ("sort-only plain bitonic", {"LZ77X_MATCH_VARIANT": "2", "LZ77X_SORT_VARIANT": "1"}), ("sort-only blocked bitonic", {"LZ77X_MATCH_VARIANT": "2", "LZ77X_SORT_VARIANT": "2"}), ("match pair-scan", {"LZ77X_MATCH_VARIANT": "3"}),
                  ("walk run 1024", {"LZ77X_WALK_RUN": "1024"}), ("walk run 4096", {"LZ77X_WALK_RUN": "4096"}), ("group 1", {"LZ77X_MATCH_GROUP": "1"}),
                  ("token v1 (global)", {"LZ77X_TOKEN_VARIANT": "1"}), ("token v2 (tile, no index)", {"LZ77X_TOKEN_VARIANT": "2"})):
    os.environ.update(env)
    L.encode(data, la, sb); se = L.last_stats()
    print(name, {k: round(v, 2) for k, v in se.items() if k.endswith("_ms")})
    for k in env: del os.environ[k]
