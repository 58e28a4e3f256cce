"""S3 (212 MB mixed, s65535 l255) encode timing on the GPU box, with the stage breakdown."""
import os, sys, time, hashlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth
n = int(os.environ.get("N", 212_000_000))
kind = os.environ.get("KIND", "mixed")
data = synth.make(kind, n, synth.SEED_S3)
d_in = torch.from_numpy(data).cuda()
sb, la = int(os.environ.get("SB", 65535)), int(os.environ.get("LA", 255))
cap = L.encode_bound(n, la, sb)
d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(int(os.environ.get("ITERS", 3))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s = L.last_stats()
    print("encode %.1f ms (%.0f MB/s)" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6),
          {k: round(s[k], 1) for k in ("k_match_ms", "k_sort_ms", "k_walk_ms", "k_token_ms", "k_tiebreak_ms", "k_chain_ms", "k_prio_ms", "k_prio_fwd_ms", "k_prio_back_ms", "k_prio_scan_ms", "prio_iters", "host_chain_ms", "host_stageb_ms")}, flush=True)
sha = hashlib.sha256(d_z[:zn].cpu().numpy().tobytes()).hexdigest()
gold = [r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json")))["full"] if (r["kind"], r["n"], r["sb"], r["la"]) == (kind, n, sb, la)]
print("zn", zn, "sha", sha[:16], "golden", (gold[0]["sha256_lz"][:16], gold[0]["sha256_lz"] == sha) if gold else None)
