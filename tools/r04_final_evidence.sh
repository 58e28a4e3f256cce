#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04b; mkdir -p $out
timeout 420 bash tools/profile_bench.sh r04 > $out/profile.log 2>&1
cp gpurun_out/profile_r04/r04_* gpurun_out/profile_r04/traffic.json $out/ 2>/dev/null
timeout 150 python tools/host_rates.py > $out/host_rates.log 2>&1; cp gpurun_out/host_rates.json $out/r04_host_rates.json
timeout 200 python tools/file_rates.py > $out/file_rates.log 2>&1; cp gpurun_out/file_rates.json $out/r04_file_rates.json
ls -la $out
