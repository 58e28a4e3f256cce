"""file -> file rates of the FILE* entry points inside one warm process (no process start): python tools/file_rates.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import lz77_amd as L
from lz77_amd import synth
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
fin, flz, fout = (os.path.join(d, "lz77x_fr." + e) for e in ("in", "lz", "out"))
out = []
for kind, n, seed in (("text", 100_000_000, synth.SEED_S1), ("text", 1_000_000_000, synth.SEED_S4), ("random", 1 << 30, synth.SEED_S2)):
    data = synth.make(kind, n, seed)
    data.tofile(fin)
    rec = {"kind": kind, "bytes": n}
    for label, env in (("default", {}), ("one_at_a_time", {"LZ77X_PIPELINE": "0"})):
        for k, v in env.items():
            os.environ[k] = v
        best_e = best_d = 1e9
        for it in range(3):
            t0 = time.perf_counter(); L.encode_path(fin, flz); t1 = time.perf_counter()
            L.decode_path(flz, fout); t2 = time.perf_counter()
            if it:
                best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
        ranges = L.last_stats()["match_launches"]
        for k in env:
            del os.environ[k]
        rec[label] = {"encode_ms": round(best_e * 1e3, 1), "decode_ms": round(best_d * 1e3, 1), "encode_GBps": round(n / best_e / 1e9, 2),
                      "decode_GBps": round(n / best_d / 1e9, 2), "decode_ranges": ranges}
    import numpy as np
    assert np.array_equal(np.fromfile(fout, dtype=np.uint8), data)
    print(rec, flush=True)
    out.append(rec)
for p in (fin, flz, fout):
    if os.path.exists(p):
        os.remove(p)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "file_rates.json"), "w"), indent=1)
