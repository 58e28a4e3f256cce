"""The run cliff of the large windows on the record: the two worst families of tools/worst_cases.py at C2 (runs of equal bytes
cut by single bytes, text with planted runs) and at C1, beside text of the same size.  python tools/run_cliff.py [bytes]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import worst_cases as W  # noqa: E402
from lz77_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
rng = np.random.default_rng(7)
out = []
for name, sb, la in (("C2 s=65535 l=255", 65535, 255), ("C1 s=4095 l=15", 4095, 15)):
    for label, data in (("text", synth.text(n, synth.SEED_S1)), ("cut_runs(run=1000, alphabet=16)", W.cut_runs(n, rng, 1000, 16)),
                        ("text_with_runs(every=2000, run=1999)", W.text_with_runs(n, rng, 2000, 1999)),
                        ("cut_runs(run=300, alphabet=2)", W.cut_runs(n, rng, 300, 2))):
        r = W.measure(data, sb, la)
        r["input"] = label
        r["geometry"] = name
        out.append(r)
        print(json.dumps(r), flush=True)
