#!/bin/bash
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_memory.py tests/test_gpu_full.py -q -x --capture=sys -p no:cacheprovider -k "large_window or hand_overs or segments or memory or s3 or s1 or s2 or carry or kat or bulk" 2>&1 | tail -2
echo "== memory"; timeout 600 python tools/mem_probe.py 2>&1 | grep "^{" | cut -c1-200
echo "== S3"; ITERS=3 timeout 300 python tools/time_c2.py 2>&1 | grep encode | tail -1 | cut -c1-60
echo "== S1"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-file-to-file --streams 1 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['stream_sha_ok'], r['roofline']['traffic'] is not None)"
