"""Scratch driver for the workgroup-wide priority recurrence (k_priow.hip); run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lz77_amd as L
import oracle_lib as O
from lz77_amd import synth

_cache = {}

def run(kind, seed, n, sb, la, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    key = (kind, seed, n, sb, la)
    if key not in _cache:
        data = synth.make(kind, n, seed)
        P, S, _ = O.stage_a(data, sb, la, tree=True)
        _cache[key] = (P, S, O.stage_b(P, S, sb))
    P, S, want = _cache[key]
    t0 = time.time()
    try:
        got, it = L.stage_priorities_device(P, S, sb)
    except Exception as e:
        print("prio %-7s n=%-8d sb=%-5d env=%s FAILED %s" % (kind, n, sb, env, e), flush=True)
        for k in (env or {}):
            os.environ.pop(k, None)
        return 1
    dt = time.time() - t0
    bad = int((got != want).sum())
    print("prio %-7s n=%-8d sb=%-5d env=%s iters=%d mismatches=%d first=%s  %.1f ms" %
          (kind, n, sb, env, it, bad, np.flatnonzero(got != want)[:4], dt * 1e3), flush=True)
    for k in (env or {}):
        os.environ.pop(k, None)
    return bad

bad = 0
W256 = {"LZ77X_PRIO_WIDE": "256"}
W1K = {"LZ77X_PRIO_WIDE": "1024"}
bad += run("text", 51, 30000, 4095, 15, W256)
bad += run("text", 51, 30000, 4095, 15, dict(W256, LZ77X_PRIO_BLOCK="4096"))
bad += run("random", 52, 200000, 1000, 10, dict(W256, LZ77X_PRIO_BLOCK="1024", LZ77X_PRIO_SCAN_GROUP="3"))
bad += run("lowent", 53, 100000, 100, 10, dict(W256, LZ77X_PRIO_BLOCK="512"))
bad += run("zeros", 0, 60000, 4095, 15, dict(W256, LZ77X_PRIO_BLOCK="4096"))
bad += run("mixed", 54, 300000, 255, 7, dict(W256, LZ77X_PRIO_BLOCK="512", LZ77X_PRIO_SCAN_GROUP="7"))
bad += run("text", 58, 12000, 1, 15, W256)
bad += run("random", 60, 9000, 3, 2, W1K)
bad += run("code", 57, 200000, 4096, 16, W1K)
bad += run("text", 0x5EED0001, 4 << 20, 4095, 15, W256)
bad += run("text", 0x5EED0001, 4 << 20, 4095, 15, W1K)
bad += run("mixed", 0x5EED0003, 4 << 20, 4095, 15, W256)
# windows above 4096: the HBM scan; 32-bit ring up to ~37 K, 18-bit codes above
bad += run("text", 61, 300000, 8191, 16)
bad += run("mixed", 62, 1 << 20, 8192, 31, {"LZ77X_PRIO_BLOCK": "16384", "LZ77X_PRIO_SCAN_GROUP": "3"})
bad += run("lowent", 63, 400000, 20000, 100, {"LZ77X_PRIO_BLOCK": "20480"})
bad += run("text", 64, 1 << 20, 40000, 255)
bad += run("text", 64, 1 << 20, 40000, 255, {"LZ77X_PRIO_SCAN_GROUP": "2", "LZ77X_PRIO_SORTCAP": "1024"})
bad += run("mixed", 65, 3 << 20, 65535, 255)
bad += run("mixed", 65, 3 << 20, 65535, 255, {"LZ77X_PRIO_SCAN_GROUP": "3", "LZ77X_PRIO_SORTCAP": "4096"})
bad += run("random", 66, 1 << 20, 65535, 255)
bad += run("records", 67, 2 << 20, 65535, 255)
bad += run("zeros", 0, 300000, 65535, 255)
bad += run("text", 68, 140000, 65535, 255)
bad += run("text", 68, 70000, 65535, 255)
bad += run("mixed", 0x5EED0003, 6 << 20, 65535, 255)
print("TOTAL mismatches", bad)
