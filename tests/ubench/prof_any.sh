# per-kernel times of one gpu_time.py run: bash tests/ubench/prof_any.sh <gpu_time.py args>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export LZ77X_ITERS=1 LZ77X_SWEEP=0
rm -rf gpurun_out/prof_any
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_any -o p -- python tests/gpu_time.py "$@" > gpurun_out/prof_any.log 2>&1
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/prof_any/p_kernel_stats.csv")))[:16]:
    print("  %-44s calls %5s total %9.2f ms avg %9.3f ms  %5s%%" % (r['Name'][:44], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage']))
PY
grep -E "^enc|^  dec" gpurun_out/prof_any.log | cut -c1-200
rm -rf gpurun_out/prof_any
