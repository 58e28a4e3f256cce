#!/bin/bash
mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decode_ranges.py -q -x --capture=sys -p no:cacheprovider -k "not hand_overs_by_rank and not long_runs and not variants and not shards_give" 2>&1 | tail -3
echo "== S1"; for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-file-to-file --streams 1 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:r[k] for k in ('value','ms_per_step','stream_sha_ok','roundtrip_ok')}, r['encode_breakdown_ms']['total_ms'], r['decode_breakdown_ms'])"; done
echo "== S3"; ITERS=3 timeout 300 python tools/time_c2.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-200
echo "== memory"; timeout 600 python tools/mem_probe.py 2>&1 | grep "^{" | cut -c1-260
echo "== shards, 8 contexts on this GPU"; LZ77X_FAKE_DEVICES=8 timeout 600 python bench.py --mode shard --gpus 8 --steps 2 --warmup 1 > gpurun_out/r06_shard_fake8.json 2> gpurun_out/r06_shard.err; python - <<'PY'
import json
try:
    r=json.loads(open("gpurun_out/r06_shard_fake8.json").read().strip().splitlines()[-1])
    print({k:r.get(k) for k in ("value","encode_ms","decode_ms","prio_iters","host_serial_ms","roundtrip_ok","stream_sha_ok")}); print(r.get("amdahl"))
except Exception as e: print("shard bench failed", e); print(open("gpurun_out/r06_shard.err").read()[-2000:])
PY
