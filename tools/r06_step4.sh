#!/bin/bash
for ch in 67108864 100663296 134217728; do echo "== S3 token chunk $ch"; LZ77X_TOKEN_CHUNK=$ch ITERS=3 timeout 300 python tools/time_c2.py 2>&1 | grep encode | tail -1 | cut -c1-120
python - <<PY
import os,sys
sys.path.insert(0,".")
os.environ["LZ77X_TOKEN_CHUNK"]="$ch"
import torch, lz77_amd as L
from lz77_amd import synth
n=212_000_000; data=synth.make("mixed", n, 77); d_in=torch.from_numpy(data).cuda(); cap=L.encode_bound(n,255,65535); d_z=torch.empty(cap,dtype=torch.uint8,device="cuda")
L.lib().lz77x_shutdown(); torch.cuda.synchronize(); f0=torch.cuda.mem_get_info()[0]
L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, 255, 65535, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize(); f1=torch.cuda.mem_get_info()[0]
print("held %.1f MB = %.1f B per input byte" % ((f0-f1)/1e6, (f0-f1)/n))
PY
done
echo "== shards"; LZ77X_FAKE_DEVICES=8 timeout 600 python bench.py --mode shard --gpus 8 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:r.get(k) for k in ('value','encode_ms','decode_ms','prio_iters','host_serial_ms','roundtrip_ok','stream_sha_ok')}); print(r['amdahl']['bound_speedup'], r['amdahl']['encode_ms_one_context'], r['amdahl']['host_serial_ms_per_gate_iteration'])"
