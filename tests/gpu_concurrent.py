"""Aggregate throughput of k concurrent streams on ONE GPU (threads of one process, each with its own
leased context): python tests/gpu_concurrent.py [k ...]"""
import os, sys, time, threading
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import torch
import lz77_amd as L
from lz77_amd import synth

n = 100_000_000
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
os.environ.setdefault("LZ77X_MAX_CONTEXTS", str(max(ks)))
datas = [torch.from_numpy(synth.text(n, synth.SEED_S1 + i)).cuda() for i in range(max(ks))]
cap = L.encode_bound(n)
outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in datas]
backs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in datas]


def step(i, reps):
    s = torch.cuda.Stream()
    for _ in range(reps):
        zn = L.encode_device(datas[i].data_ptr(), n, outs[i].data_ptr(), cap, stream=s.cuda_stream)
        L.decode_device(outs[i].data_ptr(), zn, backs[i].data_ptr(), n, stream=s.cuda_stream)


for k in ks:
    for reps in (1, 3):                                   # first round warms the k contexts up
        ts = [threading.Thread(target=step, args=(i, reps)) for i in range(k)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
    print("%d concurrent streams: %.1f MB/s aggregate encode+decode (%.1f ms per step per stream)" % (k, k * 3 * n / dt / 1e6, dt / 3 * 1e3))
    for i in range(k):
        assert torch.equal(backs[i], datas[i])
