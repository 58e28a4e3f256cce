"""Encode/decode rates of the BASELINE.json configurations on one MI355X (HBM-resident buffers, as
bench.py measures them): python tools/measure_configs.py > gpurun_out/r01_configs.json
configs[1] S1 100 MB text s4095/l15 (= bench.py), configs[2] S2 1 GiB random s4095/l15,
configs[3] S3 212 MB mixed s65535/l255."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth

out = []
for name, kind, n, sb, la, seed in (("S1 text 100 MB, s4095 l15", "text", 100_000_000, 4095, 15, synth.SEED_S1),
                                    ("S2 random 1 GiB, s4095 l15", "random", 1 << 30, 4095, 15, synth.SEED_S2),
                                    ("S3 mixed 212 MB, s65535 l255", "mixed", 212_000_000, 65535, 255, synth.SEED_S3)):
    data = synth.make(kind, n, seed)
    d_in = torch.from_numpy(data).cuda()
    cap = L.encode_bound(n, la, sb)
    d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    best = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        se = L.last_stats()
        L.decode_device(d_z.data_ptr(), zn, d_back.data_ptr(), n, st)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        sd = L.last_stats()
        if it and (best is None or t2 - t0 < best[0]):
            best = (t2 - t0, t1 - t0, t2 - t1, se, sd)
    assert torch.equal(d_back, d_in)
    tot, te, td, se, sd = best
    out.append({"config": name, "bytes": n, "ratio": round(zn / n, 4),
                "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
                "encode_MBps": round(n / te / 1e6, 1), "decode_MBps": round(n / td / 1e6, 1),
                "encode_plus_decode_MBps": round(n / tot / 1e6, 1),
                "encode_breakdown_ms": {k: round(se[k], 2) for k in ("k_match_ms", "k_sort_ms", "k_walk_ms", "k_chain_ms", "k_prio_ms", "k_prio_fwd_ms", "k_prio_back_ms", "k_prio_scan_ms",
                                                                              "k_token_ms", "k_tiebreak_ms", "host_chain_ms", "host_stageb_ms")},
                "prio_iters": se["prio_iters"],
                "decode_kernel_ms": round(sd["k_decode_ms"], 2), "roundtrip_ok": True})
    del d_in, d_z, d_back
    torch.cuda.empty_cache()
print(json.dumps({"device": torch.cuda.get_device_name(0), "configs": out}, indent=1))
