#!/bin/bash
mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x --capture=sys -p no:cacheprovider -k "shard or hand_overs_by_rank" 2>&1 | tail -2
echo "== S3"; ITERS=3 timeout 300 python tools/time_c2.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200
echo "== memory (C2 rows)"; timeout 600 python tools/mem_probe.py 2>&1 | grep "^{" | grep mixed | cut -c1-230
echo "== shards, 8 contexts on this GPU"; LZ77X_FAKE_DEVICES=8 timeout 600 python bench.py --mode shard --gpus 8 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:r.get(k) for k in ('value','encode_ms','decode_ms','prio_iters','host_serial_ms','roundtrip_ok','stream_sha_ok')}); print(r['amdahl']['bound_speedup'], r['amdahl']['encode_ms_one_context'])"
echo "== fuzz"; timeout 560 python tests/gpu_fuzz_long.py 480 7000 2>&1 | tail -3
