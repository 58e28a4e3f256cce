"""profiles/r03_prio_classes.json: the gate iteration of the priority recurrence per data class:
prio_iters, k_prio_ms and encode ms for text, mixed, lowent, records, code, zeros and random at 100 MB, s=4095 l=15,
plus the 4 MiB `mixed` case that took 106 iterations early in round 2.  python tools/prio_classes.py > out.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth

def run(kind, n, sb=4095, la=15, seed=synth.SEED_S1):
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
        s = L.last_stats()
        if it and (best is None or t1 - t0 < best[0]):
            best = (t1 - t0, s)
    L.decode_device(d_z.data_ptr(), zn, d_back.data_ptr(), n, st)
    ok = bool(torch.equal(d_back, d_in))
    t, s = best
    rec = {"kind": kind, "bytes": n, "sb": sb, "la": la, "encode_ms": round(t * 1e3, 2), "encode_MBps": round(n / t / 1e6, 1),
           "prio_iters": s["prio_iters"], "k_prio_ms": round(s["k_prio_ms"], 2), "k_prio_fwd_ms": round(s["k_prio_fwd_ms"], 2),
           "k_prio_back_ms": round(s["k_prio_back_ms"], 2), "k_prio_scan_ms": round(s["k_prio_scan_ms"], 2),
           "k_match_ms": round(s["k_match_ms"], 2), "k_token_ms": round(s["k_token_ms"], 2),
           "host_stageb_ms": round(s["host_stageb_ms"], 2), "ratio": round(zn / n, 4), "roundtrip_ok": ok}
    del d_in, d_z, d_back
    torch.cuda.empty_cache()
    return rec

out = []
n = int(os.environ.get("N", 100_000_000))
for kind in os.environ.get("KINDS", "text,mixed,lowent,records,code,zeros,random").split(","):
    out.append(run(kind, n))
    print(out[-1], file=sys.stderr, flush=True)
if "KINDS" not in os.environ:
    out.append(run("mixed", 4 << 20))
    out[-1]["note"] = "the 4 MiB case of gpurun_out/prio1.log (early round 2: 106 iterations, 141.7 ms)"
text_per_byte = out[0]["k_prio_ms"] / out[0]["bytes"]
for r in out:
    r["k_prio_vs_text_per_byte"] = round(r["k_prio_ms"] / r["bytes"] / text_per_byte, 2) if text_per_byte else None
print(json.dumps({"device": torch.cuda.get_device_name(0), "classes": out}, indent=1))
