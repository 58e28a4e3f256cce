#!/bin/bash
# Regenerates every r04 file of profiles/ in one gpurun call (results under gpurun_out/r04/): bash tools/r04_evidence.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/r04; mkdir -p $out
bash tools/profile_bench.sh r04 > $out/profile.log 2>&1
cp gpurun_out/profile_r04/r04_* gpurun_out/profile_r04/traffic.json $out/ 2>/dev/null
python tools/prio_classes.py > $out/r04_prio_classes.json 2> $out/prio_classes.err
python tools/measure_configs.py > $out/r04_configs.json 2> $out/configs.err
bash tools/prof_cmd.sh r04_c2 ITERS=1 -- python tools/time_c2.py > $out/c2.log 2>&1; cp gpurun_out/r04_c2_kernel_stats.csv $out/
python tools/mem_probe.py > $out/mem_probe.log 2>&1; cp gpurun_out/mem_probe.json $out/r04_mem_probe.json
python tools/host_rates.py > $out/host_rates.log 2>&1; cp gpurun_out/host_rates.json $out/r04_host_rates.json
bash tools/cli_trace.sh > $out/r04_cli_trace.txt 2>&1
LZ77X_FAKE_DEVICES=8 python bench.py --mode shard --gpus 8 --steps 2 --warmup 1 > $out/r04_shard_fake8.json 2> $out/shard.err
ls -la $out
