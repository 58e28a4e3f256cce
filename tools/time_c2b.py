"""fwd-sweep timing probes at C2 (results wrong under a probe): kernel times from last_stats"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth
n = int(os.environ.get("N", 212_000_000))
data = synth.make("mixed", n, synth.SEED_S3)
d_in = torch.from_numpy(data).cuda()
sb, la = 65535, 255
cap = L.encode_bound(n, la, sb)
d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
os.environ["LZ77X_PRIO_MAX_ITERS"] = "3"
for probe in ("0", "1", "3", "7", "2", "4"):
    os.environ["LZ77X_PW_PROBE"] = probe
    try:
        zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
    except Exception as e:
        print("probe", probe, "failed", e)
    s = L.last_stats()
    print("probe", probe, {k: round(s[k], 1) for k in ("k_prio_ms", "k_prio_fwd_ms", "k_prio_back_ms", "k_prio_scan_ms", "prio_iters", "total_ms")}, flush=True)
