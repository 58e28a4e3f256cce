"""device memory the library holds after an encode / a decode (hipMemGetInfo deltas): python tools/mem_probe.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lz77_amd as L
from lz77_amd import synth

def free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]

out = []
for kind, n, sb, la, env in (("text", 100_000_000, 4095, 15, {}), ("text", 400_000_000, 4095, 15, {}), ("text", 1_000_000_000, 4095, 15, {}),
                             ("text", 1_000_000_000, 4095, 15, {"LZ77X_SEGMENT": "268435456"}),
                             ("mixed", 212_000_000, 65535, 255, {}), ("mixed", 400_000_000, 65535, 255, {}), ("mixed", 200_000_000, 20000, 64, {})):
    for k, v in env.items():
        os.environ[k] = v
    L.lib().lz77x_shutdown()
    data = synth.make(kind, n, 77)
    d_in = torch.from_numpy(data).cuda()
    cap = L.encode_bound(n, la, sb)
    d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f0 = free()
    zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
    f1 = free()
    L.lib().lz77x_shutdown()
    f2 = free()
    L.decode_device(d_z.data_ptr(), zn, d_in.data_ptr(), n, st)
    f3 = free()
    L.lib().lz77x_shutdown()
    rec = {"kind": kind, "n": n, "sb": sb, "la": la, "env": env, "encode_held_MB": round((f0 - f1) / 1e6, 1), "encode_B_per_input_B": round((f0 - f1) / n, 2),
           "decode_held_MB": round((f2 - f3) / 1e6, 1), "decode_B_per_output_B": round((f2 - f3) / n, 2), "leak_MB": round((f0 - f2) / 1e6, 1)}
    print(rec, flush=True)
    out.append(rec)
    for k in env:
        del os.environ[k]
    del d_in, d_z
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mem_probe.json"), "w"), indent=1)
