"""Timing of the device priority recurrence under different block / scan-group sizes (GPU box)."""
import os, sys, json
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
for env in [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]:
    for k, v in env.items():
        os.environ[k] = v
    for it in range(2):
        zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, 15, 4095, st)
    s = L.last_stats()
    print(env, "iters", s["prio_iters"], {k: round(s[k], 2) for k in ("total_ms", "k_prio_ms", "k_prio_fwd_ms", "k_prio_back_ms", "k_prio_scan_ms", "k_match_ms", "k_token_ms")}, flush=True)
    for k in env:
        os.environ.pop(k)
