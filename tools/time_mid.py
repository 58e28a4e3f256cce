"""encode timing at a mid-size window (default s=32768 l=64 on 100 MB mixed)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth
n = int(os.environ.get("N", 100_000_000)); sb = int(os.environ.get("SB", 32768)); la = int(os.environ.get("LA", 64))
d = synth.make(os.environ.get("KIND", "mixed"), n, 1234)
di = torch.from_numpy(d).cuda(); cap = L.encode_bound(n, la, sb); dz = torch.empty(cap, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zn = L.encode_device(di.data_ptr(), n, dz.data_ptr(), cap, la, sb, st)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s = L.last_stats()
    print("sb %d la %d encode %.1f ms (%.0f MB/s)" % (sb, la, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6),
          {k: round(s[k], 1) for k in ("k_match_ms", "k_sort_ms", "k_walk_ms", "k_token_ms", "host_stageb_ms")}, flush=True)
