# per-kernel times with the token stream serialised behind the match stream:
#   bash tests/ubench/prof_serial.sh [c1|c2|both]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
which=${1:-both}
export LZ77X_SERIAL=1 LZ77X_ITERS=2 LZ77X_SWEEP=0
if [ $which != c1 ]; then rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -o c2 -- python tests/gpu_time.py mixed 212000000 65535 255 > gpurun_out/prof_c2.log 2>&1; fi
if [ $which != c2 ]; then rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c1 -o c1 -- python tests/gpu_time.py > gpurun_out/prof_c1.log 2>&1; fi
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/prof_c[12]/*kernel_stats.csv")):
    print(f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  %-40s calls %4s total %9.2f ms avg %9.3f ms  %5s%%" % (r['Name'][:40], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage']))
PY
find gpurun_out/prof_c2 gpurun_out/prof_c1 -name "*kernel_trace.csv" -delete 2>/dev/null
