"""S2 (1 GiB random, s4095 l15) encode timing with the recurrence's breakdown; KIND/N override."""
import os, sys, time, hashlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth
n = int(os.environ.get("N", 1 << 30)); kind = os.environ.get("KIND", "random")
seed = {"random": synth.SEED_S2, "text": synth.SEED_S4}.get(kind, synth.SEED_S2)
d = synth.make(kind, n, seed)
di = torch.from_numpy(d).cuda(); cap = L.encode_bound(n, 15, 4095); dz = torch.empty(cap, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(int(os.environ.get("ITERS", 3))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zn = L.encode_device(di.data_ptr(), n, dz.data_ptr(), cap, 15, 4095, st)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s = L.last_stats()
    print("%s %d encode %.1f ms (%.0f MB/s)" % (kind, n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6),
          {k: round(s[k], 1) for k in ("k_match_ms", "k_prio_ms", "k_prio_fwd_ms", "k_prio_back_ms", "k_prio_scan_ms", "k_token_ms", "k_tiebreak_ms", "k_chain_ms")}, s["prio_iters"], flush=True)
sha = hashlib.sha256(dz[:zn].cpu().numpy().tobytes()).hexdigest()
gold = [r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json")))["full"] if (r["kind"], r["n"], r["sb"], r["la"]) == (kind, n, 4095, 15)]
print("zn", zn, "sha", sha[:16], "golden", (gold[0]["sha256_lz"][:16], gold[0]["sha256_lz"] == sha) if gold else None)
