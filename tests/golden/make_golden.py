#!/usr/bin/env python3
"""Regenerate tests/golden/ from the COMPILED REFERENCE (oracle/_ref/lz77_ref).

Run in the build container only (needs /root/reference via `make -C oracle ref`):

    python tests/golden/make_golden.py

Everything written here is DATA: inputs are produced by lz77_amd/synth.py (splitmix64
seeded), outputs are what the reference binary emitted for them.  No reference source
text is stored.  Files:

    golden.json        KATs (hex), bulk digests, size-grid digests
    small_*.bin/.lz    a few <=64 KiB inputs with the reference's compressed stream
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
from lz77_amd import synth  # noqa: E402


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    G = {"about": "outputs of cstdvd/lz77 (gcc -O2, -lm) on synthetic inputs; see make_golden.py"}

    # ---- known-answer vectors --------------------------------
    kat_inputs = {
        "abra": b"abracadabra abracadabra abracadabra",
        "a40": b"a" * 40,
        "a1": b"a",
        "empty": b"",
        "ab300": b"ab" * 150,
        "bytes256x3": bytes(range(256)) * 3,
    }
    geoms = [(4095, 15), (65535, 255), (1000, 10), (255, 7), (16, 4), (7, 3), (4096, 16), (3, 200)]
    G["kat"] = []
    for name, data in kat_inputs.items():
        for sb, la in geoms:
            z = O.ref_encode(data, sb, la)
            G["kat"].append({"name": name, "input_hex": data.hex(), "sb": sb, "la": la, "lz_hex": z.hex(),
                             "decoded_hex": O.ref_decode(z).hex()})

    # ---- bulk digests ------------------------------------------------------------------
    # tier "cpu": small enough for the oracle in the CPU suite; tier "gpu": GPU parity only
    bulk = [
        ("text", 0x5EED0001, 2 << 20, 4095, 15, "cpu"),
        ("random", 0x5EED0002, 1 << 20, 4095, 15, "cpu"),
        ("lowent", 7, 1 << 19, 4095, 15, "cpu"),
        ("code", 13, 1 << 20, 4095, 15, "cpu"),
        ("records", 11, 1 << 20, 4095, 15, "cpu"),
        ("zeros", 0, 40000, 4095, 15, "cpu"),
        ("mixed", 0x5EED0003, 3 << 20, 65535, 255, "cpu"),
        ("text", 0x5EED0001, 1 << 20, 65535, 255, "cpu"),
        ("random", 0x5EED0002, 1 << 19, 65535, 255, "cpu"),
        ("text", 21, 1 << 20, 1000, 10, "cpu"),
        ("mixed", 22, 1 << 20, 255, 7, "cpu"),
        ("code", 23, 1 << 19, 4096, 16, "cpu"),
        ("text", 24, 1 << 19, 100, 200, "cpu"),
        ("lowent", 25, 1 << 18, 5, 3, "cpu"),
        ("text", 0x5EED0001, 16 << 20, 4095, 15, "gpu"),
        ("random", 0x5EED0002, 16 << 20, 4095, 15, "gpu"),
        ("mixed", 0x5EED0003, 16 << 20, 4095, 15, "gpu"),
        ("mixed", 0x5EED0003, 8 << 20, 65535, 255, "gpu"),
    ]
    G["bulk"] = []
    for kind, seed, n, sb, la, tier in bulk:
        data = synth.make(kind, n, seed)
        z = O.ref_encode(data, sb, la)
        T = O.token_bits(sb, la)
        G["bulk"].append({"kind": kind, "seed": seed, "n": n, "sb": sb, "la": la, "tier": tier,
                          "zn": len(z), "ntok": (len(z) * 8 - 32) // T,
                          "sha256_in": sha(data), "sha256_lz": sha(z)})
        print("bulk", kind, n, sb, la, "ratio %.4f" % (len(z) / max(n, 1)), flush=True)

    # ---- size-boundary grid -----------------------------
    G["grid"] = []
    for sb, la in ((10, 4), (7, 3), (4095, 15), (64, 16)):
        W = 3 * sb + la
        sizes = sorted({0, 1, 2, la - 1, la, la + 1, sb - 1, sb, sb + 1, sb + la, W - 1, W, W + 1, 2 * sb,
                        W + 2 * sb - 1, W + 2 * sb, W + 2 * sb + 1, W + 4 * sb, W + 4 * sb + 1, 5 * W + 3})
        for kind, seed in (("lowent", 31), ("zeros", 0), ("random", 32)):
            for n in sizes:
                data = synth.make(kind, n, seed)
                z = O.ref_encode(data, sb, la)
                G["grid"].append({"kind": kind, "seed": seed, "n": n, "sb": sb, "la": la,
                                  "zn": len(z), "sha256_lz": sha(z)})

    # ---- small files with streams (decoder fixtures) ---------------------------------
    G["small"] = []
    for kind, seed, n, sb, la in (("text", 41, 65536, 4095, 15), ("random", 42, 20000, 4095, 15),
                                 ("lowent", 43, 50000, 4095, 15), ("mixed", 44, 65536, 1000, 10),
                                 ("text", 45, 40000, 65535, 255), ("code", 46, 30000, 255, 7)):
        data = synth.make(kind, n, seed)
        z = O.ref_encode(data, sb, la)
        stem = "small_%s_%d_%d" % (kind, sb, la)
        np.asarray(data).tofile(os.path.join(HERE, stem + ".bin"))
        with open(os.path.join(HERE, stem + ".lz"), "wb") as f:
            f.write(z)
        G["small"].append({"stem": stem, "kind": kind, "seed": seed, "n": n, "sb": sb, "la": la,
                           "sha256_in": sha(data), "sha256_lz": sha(z)})

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(G, f, indent=0, sort_keys=True)
    print("wrote golden.json:", len(G["kat"]), "kat,", len(G["bulk"]), "bulk,", len(G["grid"]), "grid")


if __name__ == "__main__":
    main()
