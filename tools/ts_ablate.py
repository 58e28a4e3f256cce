"""Timing ablations of k_tokens_sorted (GPU box): LZ77X_TS_ABLATE = 1 setup only, 2 + run search, 3 + batch scans (no members)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth
n = int(os.environ.get("N", 100_000_000))
data = synth.make(os.environ.get("KIND", "text"), n, synth.SEED_S1)
d_in = torch.from_numpy(data).cuda()
cap = L.encode_bound(n, 15, 4095)
d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for ab in (os.environ.get("ABL", "0,1,2,3")).split(","):
    os.environ["LZ77X_TS_ABLATE"] = ab
    for it in range(2):
        L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, 15, 4095, st)
    s = L.last_stats()
    print("ablate", ab, {k: round(s[k], 2) for k in ("k_tiebreak_ms", "k_token_ms")}, flush=True)
