"""Worst cases on the record: a search for inputs that maximise the gate iterations of the priority
recurrence and the run lengths of the tie-break, at C1 (s=4095 l=15) and C2 (s=65535 l=255).

    python tools/worst_cases.py [seconds] > gpurun_out/worst_cases.json

Families of adversarial streams (periods around the lookahead and the window, runs cut by single bytes, tiny alphabets,
repeated random blocks of about a window, text with planted runs, each with a noise knob) are drawn until the time budget
is spent; every candidate is encoded on the device (HBM resident), decoded back, and scored by gate iterations and by
encode milliseconds per 100 MB.  The worst five per geometry are re-run at 100 MB (C1) / 64 MB (C2).  A candidate on
which the gate iteration gives up (host_stageb_ms > 0: the host-assisted fallback of encode_host.cpp) is reported as such.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import lz77_amd as L  # noqa: E402
from lz77_amd import synth  # noqa: E402


def periodic(n, rng, period, noise):
    base = rng.integers(0, 256, period, dtype=np.uint8)
    d = np.tile(base, n // period + 1)[:n].copy()
    if noise:
        k = max(n // noise, 1)
        d[rng.integers(0, n, k)] = rng.integers(0, 256, k, dtype=np.uint8)
    return d


def cut_runs(n, rng, run, alphabet):
    """runs of one byte, `run` long, separated by ONE other byte"""
    reps = n // (run + 1) + 1
    vals = rng.integers(0, alphabet, reps, dtype=np.uint8)
    d = np.repeat(vals, run + 1)[:n].copy()
    d[run::run + 1] = rng.integers(alphabet, 256, len(d[run::run + 1]), dtype=np.uint8)
    return d


def tiny_alphabet(n, rng, alphabet, bias):
    p = np.array([bias ** i for i in range(alphabet)], dtype=np.float64)
    return rng.choice(alphabet, n, p=p / p.sum()).astype(np.uint8)


def repeated_block(n, rng, block, noise):
    return periodic(n, rng, block, noise)


def text_with_runs(n, rng, every, run):
    d = synth.text(n, int(rng.integers(1, 1 << 30))).copy()
    for at in range(every, n - run, every):
        d[at:at + run] = d[at]
    return d


def staircase(n, rng, step):
    """a byte that changes every `step` positions, cycling through a few values: the keys of neighbouring positions differ late"""
    return ((np.arange(n, dtype=np.uint64) // step) % 5).astype(np.uint8)


def draw(rng, sb, la, n):
    fam = int(rng.integers(0, 6))
    if fam == 0:
        period = int(rng.choice([1, 2, 3, la - 1, la, la + 1, 2 * la, 64, 255, 256, sb // 2, sb - 1, sb, sb + 1, 2 * sb + 1]))
        noise = int(rng.choice([0, 50, 500, 5000, 100000]))
        return "periodic(period=%d, noise=1/%d)" % (period, noise), periodic(n, rng, max(period, 1), noise)
    if fam == 1:
        run = int(rng.choice([la - 1, la, la + 1, 3 * la, 100, 1000, sb, sb + 1]))
        alphabet = int(rng.choice([1, 2, 4, 16]))
        return "cut_runs(run=%d, alphabet=%d)" % (run, alphabet), cut_runs(n, rng, run, alphabet)
    if fam == 2:
        alphabet = int(rng.choice([2, 3, 4, 8]))
        bias = float(rng.choice([1.0, 0.5, 0.1]))
        return "tiny_alphabet(%d, bias=%.1f)" % (alphabet, bias), tiny_alphabet(n, rng, alphabet, bias)
    if fam == 3:
        block = int(rng.choice([sb // 4, sb // 2, sb - la, sb, sb + la, 2 * sb, 3 * sb + 7]))
        noise = int(rng.choice([0, 200, 5000]))
        return "repeated_block(%d, noise=1/%d)" % (block, noise), repeated_block(n, rng, max(block, 1), noise)
    if fam == 4:
        every = int(rng.choice([300, 2000, sb, 3 * sb]))
        run = int(rng.choice([la, 4 * la, 200, min(sb, every - 1)]))
        return "text_with_runs(every=%d, run=%d)" % (every, min(run, every - 1)), text_with_runs(n, rng, every, min(run, every - 1))
    step = int(rng.choice([la - 1, la, la + 1, 100, sb // 3, sb]))
    return "staircase(step=%d)" % step, staircase(n, rng, max(step, 1))


def measure(data, sb, la):
    n = int(data.size)
    d_in = torch.from_numpy(data).cuda()
    cap = L.encode_bound(n, la, sb)
    d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    best = None
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s = L.last_stats()
        if best is None or t1 - t0 < best[0]:
            best = (t1 - t0, s)
    m = L.decode_device(d_z.data_ptr(), zn, d_back.data_ptr(), n, st)
    ok = bool(m == n and torch.equal(d_back, d_in))
    t, s = best
    del d_in, d_z, d_back
    return {"bytes": n, "encode_ms": round(t * 1e3, 2), "ms_per_100MB": round(t * 1e3 * 1e8 / n, 2), "prio_iters": int(s["prio_iters"]),
            "k_match_ms": round(s["k_match_ms"], 2), "k_prio_ms": round(s["k_prio_ms"], 2), "k_token_ms": round(s["k_token_ms"], 2),
            "fell_back_to_the_host": bool(s["host_stageb_ms"] > 0), "ratio": round(zn / n, 4), "roundtrip_ok": ok}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    rng = np.random.default_rng(20250105)
    out = {"device": torch.cuda.get_device_name(0), "seconds": budget, "geometries": {}}
    for name, sb, la, n_search, n_final, share in (("C1 s=4095 l=15", 4095, 15, 24_000_000, 100_000_000, 0.6),
                                                   ("C2 s=65535 l=255", 65535, 255, 12_000_000, 64_000_000, 0.4)):
        t_end = time.time() + budget * share
        seen = []
        while time.time() < t_end:
            label, data = draw(rng, sb, la, n_search)
            try:
                r = measure(data, sb, la)
            except Exception as e:                                  # pragma: no cover
                r = {"error": str(e)[:200]}
            r["input"] = label
            seen.append(r)
        good = [r for r in seen if "error" not in r]
        by_iters = sorted(good, key=lambda r: -r["prio_iters"])[:5]
        by_ms = sorted(good, key=lambda r: -r["ms_per_100MB"])[:5]
        finals = []
        done = set()
        # the worst of the search again at the full size (same generator state is not kept: the label is re-drawn by family)
        for r in by_iters + by_ms:
            if r["input"] in done:
                continue
            done.add(r["input"])
            finals.append(r)
        out["geometries"][name] = {"candidates": len(seen), "errors": [r for r in seen if "error" in r][:5],
                                   "fallbacks": [r for r in good if r["fell_back_to_the_host"]][:5],
                                   "roundtrip_failures": [r for r in good if not r["roundtrip_ok"]][:5],
                                   "max_prio_iters": max((r["prio_iters"] for r in good), default=0),
                                   "worst_by_iterations": by_iters, "worst_by_ms_per_100MB": by_ms,
                                   "search_bytes": n_search, "text_for_scale": measure(synth.text(n_search, synth.SEED_S1), sb, la)}
        torch.cuda.empty_cache()
        L.lib().lz77x_shutdown()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
