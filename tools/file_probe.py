"""1 GB text: device-resident and file -> file encode at several segment sizes, with the library's trace: python tools/file_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth
n = int(os.environ.get("N", 1_000_000_000))
data = synth.make("text", n, synth.SEED_S4)
d = "/dev/shm"
fin, flz, fout = (os.path.join(d, "lz77x_fp." + e) for e in ("in", "lz", "out"))
data.tofile(fin)
d_in = torch.from_numpy(data).cuda()
cap = L.encode_bound(n)
d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for seg in os.environ.get("SEGS", "0,268435456,134217728,67108864").split(","):
    if int(seg):
        os.environ["LZ77X_SEGMENT"] = seg
    else:
        os.environ.pop("LZ77X_SEGMENT", None)
    for _ in range(2):
        t0 = time.perf_counter(); L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, -1, -1, st); torch.cuda.synchronize(); t1 = time.perf_counter()
    dev_ms = (t1 - t0) * 1e3
    best = 1e9
    for it in range(3):
        if it == 2 and os.environ.get("TRACE"):
            os.environ["LZ77X_TRACE"] = "1"
        t0 = time.perf_counter(); L.encode_path(fin, flz); t1 = time.perf_counter()
        os.environ.pop("LZ77X_TRACE", None)
        best = min(best, t1 - t0)
    bestd = 1e9
    for it in range(2):
        t0 = time.perf_counter(); L.decode_path(flz, fout); t1 = time.perf_counter()
        bestd = min(bestd, t1 - t0)
    print("segment", seg, "device-resident %.1f ms" % dev_ms, "file->file encode %.1f ms (%.2f GB/s)" % (best * 1e3, n / best / 1e9),
          "decode %.1f ms (%.2f GB/s)" % (bestd * 1e3, n / bestd / 1e9), flush=True)
import numpy as np
assert np.array_equal(np.fromfile(fout, dtype=np.uint8), data)
for p in (fin, flz, fout):
    os.remove(p)
