"""The error front of the gate iteration: LZ77X_PRIO_TRACE=1 python tools/prio_front_probe.py -- a random block of
about a window, repeated: flips per iteration, the first open block, when the library gives up and what the encode costs."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lz77_amd as L
rng = np.random.default_rng(1)
for period, n in ((4096, 24_000_000), (4095, 24_000_000), (8190, 24_000_000), (300, 24_000_000)):
    d = np.tile(rng.integers(0, 256, period, dtype=np.uint8), n // period + 1)[:n].copy()
    x = torch.from_numpy(d).cuda(); cap = L.encode_bound(n); z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        zn = L.encode_device(x.data_ptr(), n, z.data_ptr(), cap, 15, 4095, st); torch.cuda.synchronize()
        t1 = time.perf_counter()
    s = L.last_stats()
    print("period", period, "ms", round((t1 - t0) * 1e3, 1), {k: round(s[k], 1) for k in ("prio_iters", "k_prio_ms", "host_stageb_ms", "k_match_ms", "k_token_ms", "total_ms")}, flush=True)
