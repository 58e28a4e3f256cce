#!/bin/bash
mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --capture=sys -p no:cacheprovider -k "hand_overs_by_rank or large_window_short or s3" 2>&1 | tail -3
echo "== S3"; ITERS=3 timeout 300 python tools/time_c2.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-230
echo "== kernel stats"; bash tools/prof_cmd.sh r06_c2b ITERS=1 -- python tools/time_c2.py | grep -i "tokens\|k_hr\|k_sx\|k_xfer"
for p in 1 4; do echo "== RANK_PROBE=$p"; LZ77X_RANK_PROBE=$p ITERS=2 python tools/time_c2.py 2>&1 | grep encode | tail -1 | cut -c1-160; done
