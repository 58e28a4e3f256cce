"""match-stage split per data class (100 MB, C1): python tools/match_split.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth
n = int(os.environ.get("N", 100_000_000))
for kind in os.environ.get("KINDS", "text,lowent,zeros,mixed,records,random").split(","):
    data = synth.make(kind, n, synth.SEED_S1)
    d_in = torch.from_numpy(data).cuda()
    cap = L.encode_bound(n)
    d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, -1, -1, st)
    s = L.last_stats()
    print(kind, {k: round(s[k], 2) for k in ("total_ms", "k_match_ms", "k_sort_ms", "k_sort_chunks_ms", "k_walk_ms", "k_token_ms", "k_tiebreak_ms", "k_prio_ms", "k_chain_ms")},
          "final+export", round(s["k_match_ms"] - s["k_sort_ms"] - s["k_walk_ms"], 2), flush=True)
    del d_in, d_z
