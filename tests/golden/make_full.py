#!/usr/bin/env python3
"""Pin the BASELINE-size streams to the COMPILED REFERENCE (oracle/_ref/lz77_ref).

Build container only (needs /root/reference via `make -C oracle ref`):

    python tests/golden/make_full.py [job ...]        # default: every job, 4 at a time

Writes tests/golden/golden_full.json: for each full-size configuration of BASELINE.json the input
size, the reference's stream size / token count and sha256 of input and stream.  A round trip cannot
see a wrong tie-break offset (any valid offset decodes), so bench.py, tools/measure_configs.py and
the `-m gpu` full-size tests compare the digest of the device stream with these.  DATA only: inputs
come from lz77_amd/synth.py (splitmix64 seeded), outputs are what the reference binary emitted.
"""
import hashlib
import json
import multiprocessing as mp
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OUT = os.path.join(HERE, "golden_full.json")

# name -> (kind, n, seed, sb, la).  S1..S4 are SURVEY 8d / BASELINE.json configs[1..4]; S1r<k> are the
# per-rank streams of bench.py at N > 1 (seed + rank); S5 and S6 cross 4 GiB (device-side segmentation).
JOBS = {
    "S1": ("text", 100_000_000, 0x5EED0001, 4095, 15),
    "S3": ("mixed", 212_000_000, 0x5EED0003, 65535, 255),
    "S2": ("random", 1 << 30, 0x5EED0002, 4095, 15),
    "S4": ("text", 1_000_000_000, 0x5EED0004, 4095, 15),
    "S5": ("text", 5 * (1 << 30), 0x5EED0005, 4095, 15),
    # crosses 4 GiB like S5 but generates in seconds (raw splitmix64): the segment carry in the DEFAULT gpu suite
    "S6": ("random", 4_400_000_000, 0x5EED0006, 4095, 15),
}
for _r in range(1, 8):
    JOBS["S1r%d" % _r] = ("text", 100_000_000, 0x5EED0001 + _r, 4095, 15)


def sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def run(name):
    import oracle_lib as O
    from lz77_amd import synth
    kind, n, seed, sb, la = JOBS[name]
    t0 = time.time()
    tmp = "/tmp"
    fin = os.path.join(tmp, "lz77full_%s.in" % name)
    flz = os.path.join(tmp, "lz77full_%s.lz" % name)
    h = hashlib.sha256()
    with open(fin, "wb") as f:                      # piecewise for the 5 GiB job: text() is chunk-independent per call only
        data = synth.make(kind, n, seed)
        h.update(memoryview(data))
        data.tofile(f)
        del data
    try:
        subprocess.check_call([O.REF_BIN, "-c", "-i", fin, "-o", flz, "-s", str(sb), "-l", str(la)])
        zn = os.path.getsize(flz)
        T = O.token_bits(sb, la)
        rec = {"name": name, "kind": kind, "seed": seed, "n": n, "sb": sb, "la": la, "zn": zn,
               "ntok": (zn * 8 - 32) // T, "sha256_in": h.hexdigest(), "sha256_lz": sha_file(flz),
               "ref_encode_s": None}
    finally:
        for p in (fin, flz):
            if os.path.exists(p):
                os.unlink(p)
    rec["ref_encode_s"] = round(time.time() - t0, 1)
    print("done", name, rec, flush=True)
    return rec


def main():
    import oracle_lib as O
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    names = sys.argv[1:] or list(JOBS)
    have = {}
    if os.path.exists(OUT):
        have = {r["name"]: r for r in json.load(open(OUT))["full"]}
    with mp.Pool(4, maxtasksperchild=1) as pool:
        for rec in pool.imap_unordered(run, names):
            have[rec["name"]] = rec
            with open(OUT, "w") as f:
                json.dump({"about": "sha256 of cstdvd/lz77's stream (gcc -O2, -lm) on the full-size synthetic "
                                    "inputs of BASELINE.json; see make_full.py",
                           "full": [have[k] for k in sorted(have)]}, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
