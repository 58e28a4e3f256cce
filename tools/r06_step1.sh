#!/bin/bash
# round 6, first measurement of the hand-overs by rank (C2), the aliased stage buffers and the per-shard threads
mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --capture=sys -p no:cacheprovider -k "hand_overs_by_rank or large_window or shard or segments or s3 or carry" > gpurun_out/r06_s1_tests.txt 2> gpurun_out/r06_s1_tests.err
echo "tests rc=$?"; tail -5 gpurun_out/r06_s1_tests.txt; grep -v amdgpu.ids gpurun_out/r06_s1_tests.err | head -5
echo "== S3 with the rank index"; ITERS=3 timeout 300 python tools/time_c2.py 2>&1 | tail -5
echo "== S3, the walk of round 5 (variants build)"; LZ77X_NO_RANK_INDEX=1 ITERS=2 timeout 300 python tools/time_c2.py 2>&1 | tail -4
echo "== scratch held by the runtime"; ./tools/scratch_hold_probe
echo "== S1"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-file-to-file --streams 1 > gpurun_out/r06_s1_bench.json 2> gpurun_out/r06_s1_bench.err; python - <<'PY'
import json
try:
    r=json.loads(open("gpurun_out/r06_s1_bench.json").read().strip().splitlines()[-1])
    print({k:r[k] for k in ("value","ms_per_step","stream_sha_ok","roundtrip_ok")}, r["encode_breakdown_ms"])
except Exception as e: print("bench failed", e); print(open("gpurun_out/r06_s1_bench.err").read()[-2000:])
PY
echo "== memory"; timeout 600 python tools/mem_probe.py > gpurun_out/r06_s1_mem.log 2>&1; python - <<'PY'
import json
try:
    m=json.load(open("gpurun_out/mem_probe.json")); print(json.dumps(m)[:3000])
except Exception as e: print("mem probe failed", e); print(open("gpurun_out/r06_s1_mem.log").read()[-1500:])
PY
