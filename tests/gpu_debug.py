"""Stage-by-stage triage on the GPU box: python tests/gpu_debug.py  (prints which stage diverges)"""
import os
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np

import lz77_amd as L
import oracle_lib as O
from lz77_amd import synth


def first_diff(a, b):
    m = min(len(a), len(b))
    a = np.frombuffer(bytes(a[:m]), dtype=np.uint8) if not isinstance(a, np.ndarray) else a[:m]
    b = np.frombuffer(bytes(b[:m]), dtype=np.uint8) if not isinstance(b, np.ndarray) else b[:m]
    d = np.nonzero(a != b)[0]
    return (int(d[0]), int(d.size)) if d.size else None


def run(kind, seed, n, sb, la):
    tag = "%s n=%d sb=%d la=%d" % (kind, n, sb, la)
    data = synth.make(kind, n, seed)
    ok = True
    try:
        t = time.time()
        P, S, _ = O.stage_a(data, sb, la, tree=True)
        gP, gS = L.stage_neighbours(data, la, sb)
        nx = max(n - sb, 0)
        dP, dS = first_diff(gP[:nx], P[:nx]), first_diff(gS[:nx], S[:nx])
        if dP or dS:
            ok = False
            print("  [A] neighbours differ", tag, "P", dP, "S", dS)
            for name, d, g, o in (("P", dP, gP, P), ("S", dS, gS, S)):
                if d:
                    i = d[0]
                    print("     %s first at %d: gpu %s oracle %s" % (name, i, g[i:i + 6], o[i:i + 6]))
        ml = L.stage_maxlen(data, la, sb)
        z = O.encode_bst(data, sb, la)
        _, _, off, ln, nx_ = O.tokens(z)
        chain = np.concatenate([[0], np.cumsum(ln + 1)[:-1]]).astype(np.int64) if ln.size else np.zeros(0, np.int64)
        bad = np.nonzero(ml[chain] != ln.astype(np.uint8))[0]
        if bad.size:
            ok = False
            k = bad[0]
            print("  [M] maxlen differs on chain", tag, "count", bad.size, "first tok", k, "pos", chain[k], "gpu", ml[chain[k]], "want", ln[k])
        gz = L.encode(data, la, sb)
        if gz != z:
            ok = False
            print("  [E] stream differs", tag, "len", len(gz), len(z), "first", first_diff(gz, z))
            if len(gz) == len(z):
                _, _, goff, gln, gnx = O.tokens(gz)
                print("     tokens: off diff %d len diff %d next diff %d of %d" % (
                    int((goff != off).sum()), int((gln != ln).sum()), int((gnx != nx_).sum()), off.size))
        if sb & (sb - 1):
            back = L.decode(z)
            if back != data.tobytes():
                ok = False
                print("  [D] decode differs", tag, len(back), n, first_diff(back, data.tobytes()))
        print("%s %s  (%.1fs) %s" % ("OK  " if ok else "FAIL", tag, time.time() - t, L.last_stats() if ok else ""))
    except Exception:
        traceback.print_exc()
        print("EXC ", tag)
        ok = False
    return ok


if __name__ == "__main__":
    print(L.lib().lz77x_version(), "devices", L.lib().lz77x_device_count())
    cases = [("text", 1, 35, 4095, 15), ("text", 1, 5000, 4095, 15), ("text", 51, 30000, 4095, 15),
             ("random", 52, 20000, 4095, 15), ("lowent", 53, 20000, 1000, 10), ("mixed", 54, 30000, 255, 7),
             ("text", 55, 3000, 100, 200), ("lowent", 56, 4000, 5, 3), ("zeros", 0, 9000, 4095, 15),
             ("code", 57, 20000, 4096, 16), ("text", 58, 12000, 1, 15), ("text", 59, 200000, 4095, 15),
             ("mixed", 62, 140000, 65535, 255), ("lowent", 63, 30000, 8192, 16), ("text", 1, 2 << 20, 4095, 15)]
    res = [run(*c) for c in cases]
    print("passed %d / %d" % (sum(res), len(res)))
    sys.exit(0 if all(res) else 1)
