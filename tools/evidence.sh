#!/bin/bash
# Regenerates EVERY <tag> file of profiles/ from ONE gpurun call, then the table in profiles/README.md from those files
# (numbers generated, not typed):
#     gpurun --timeout 2400 -- 'bash tools/evidence.sh'        # results under gpurun_out/<tag>/
#     cp gpurun_out/<tag>/<tag>_* gpurun_out/<tag>/traffic.json profiles/ && python tools/profiles_readme.py
# PMC counters are collected in passes of their own with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3).
tag=${TAG:-r06}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-configs --no-file-to-file --streams 1"

# 0. what an instruction costs a SIMD (tests/ubench): prices the instruction-issue view of the roofline
./tests/ubench/valu_rates.bin  > $out/${tag}_valu_rates.txt 2>&1
./tests/ubench/issue_rates.bin > $out/${tag}_issue_rates.txt 2>&1

# 1. the bench line the driver will see (with the CPU baseline, S2 / S3, file to file)
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/bench.err

# 2. kernel durations of the same workload + the bench line printed inside the profiled run
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B --steps 3 --warmup 1 > $out/trace.log 2>&1
cp $out/trace/t_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
grep -o '{"metric.*' $out/trace.log > $out/${tag}_bench_under_rocprof.json

# 3. PMC passes (one encode + one decode each)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -o p -- $B --steps 1 --warmup 0 > $out/pmc_$name.log 2>&1
done
python tools/pmc_summary.py $out $tag > $out/pmc_summary.log 2>&1

# 4. where the tie-break's instructions go (variants build: the kernel leaves after a phase).  k_tokens_sorted did not change in
#    round 6 (profiles/r05_ts_probe.txt stands): EVIDENCE_TS_PROBE=1 reruns it
if [ -n "$EVIDENCE_TS_PROBE" ]; then bash tools/ts_probe.sh > $out/${tag}_ts_probe.txt 2>&1; fi

# 5. the other configurations, data classes, large-window kernels
python tools/measure_configs.py > $out/${tag}_configs.json 2> $out/configs.err
python tools/prio_classes.py > $out/${tag}_prio_classes.json 2> $out/prio_classes.err
bash tools/prof_cmd.sh ${tag}_c2 ITERS=1 -- python tools/time_c2.py > $out/c2.log 2>&1; cp gpurun_out/${tag}_c2_kernel_stats.csv $out/

bash tools/c2_pmc.sh > $out/c2_pmc.log 2>&1; cp gpurun_out/${tag}c2/${tag}c2_bench_pmc_summary.csv $out/${tag}_c2_pmc_summary.csv
# the large-window tie-break by parts (variants build, timing only): all / no deferred tokens / only the bucket tokens deferred / the walk of round 5
for v in "LZ77X_RANK_PROBE=0" "LZ77X_RANK_PROBE=1" "LZ77X_RANK_PROBE=4" "LZ77X_NO_RANK_INDEX=1"; do
  echo "== $v"; env $v ITERS=2 python tools/time_c2.py 2>&1 | grep encode | tail -1
done > $out/${tag}_c2_rank_probe.txt 2>&1

# 6. host paths: buffers, files, CLI, memory
python tools/host_rates.py > $out/host_rates.log 2>&1; cp gpurun_out/host_rates.json $out/${tag}_host_rates.json
timeout 300 python tools/file_rates.py > $out/file_rates.log 2>&1; cp gpurun_out/file_rates.json $out/${tag}_file_rates.json
python tools/mem_probe.py > $out/mem_probe.log 2>&1; cp gpurun_out/mem_probe.json $out/${tag}_mem_probe.json
./tools/scratch_hold_probe > $out/${tag}_scratch_hold.txt 2>&1      # what the runtime keeps of a queue's scratch: the "leak" of mem_probe
bash tools/cli_trace.sh > $out/${tag}_cli_trace.txt 2>&1
python tools/rss_probe.py > $out/${tag}_rss_probe.txt 2>&1

# 7. one stream over several contexts SHARING this GPU (no physical scaling is measured or claimed) and the driver's N > 1 launch line
LZ77X_FAKE_DEVICES=8 python bench.py --mode shard --gpus 8 --steps 2 --warmup 1 > $out/${tag}_shard_fake8.json 2> $out/shard.err
LZ77_BENCH_BACKEND=gloo LZ77X_FAKE_DEVICES=2 python bench.py --gpus 2 --steps 3 --warmup 1 --no-shard-record > $out/${tag}_n2_gloo_one_gpu.json 2> $out/n2.err

# 8. the worst cases on the record, the run cliff of the large windows, an open-ended fuzz (EVIDENCE_QUICK=1 skips them: ~9 minutes)
if [ -z "$EVIDENCE_QUICK" ]; then
  timeout 300 python tools/worst_cases.py 150 > $out/${tag}_worst_cases.json 2> $out/worst.err
  timeout 200 python tools/run_cliff.py 12000000 2>/dev/null | grep "^{" > $out/${tag}_run_cliff.jsonl
  timeout 260 python tests/gpu_fuzz_long.py 200 6000 2>/dev/null | tail -1 > $out/${tag}_fuzz_long.txt
fi

rm -rf $out/trace $out/pmc_*/
ls -la $out
