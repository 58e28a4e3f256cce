#!/bin/bash
# large windows: encode time and device memory held against the token chunk size (positions per hand-over index / tie-break launch)
for ch in ${CHUNKS:-41943040 50331648 67108864}; do echo "== token chunk $ch"; LZ77X_TOKEN_CHUNK=$ch ITERS=3 timeout 300 python tools/time_c2.py 2>&1 | grep encode | tail -1 | cut -c1-60
LZ77X_TOKEN_CHUNK=$ch python - <<'PY'
import os,sys
sys.path.insert(0,".")
import torch, lz77_amd as L
from lz77_amd import synth
n=212_000_000; data=synth.make("mixed", n, 77); d_in=torch.from_numpy(data).cuda(); cap=L.encode_bound(n,255,65535); d_z=torch.empty(cap,dtype=torch.uint8,device="cuda")
L.lib().lz77x_shutdown(); torch.cuda.synchronize(); f0=torch.cuda.mem_get_info()[0]
L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, 255, 65535, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize(); f1=torch.cuda.mem_get_info()[0]
print("held %.1f MB = %.1f B per input byte" % ((f0-f1)/1e6, (f0-f1)/n))
PY
done
