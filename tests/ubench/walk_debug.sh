# ablation of k_walk_big: LZ77X_WALK_DEBUG bit0 = skip the bitmap fill, bit1 = skip the steps (results are wrong, timing only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export LZ77X_SERIAL=1 LZ77X_ITERS=1 LZ77X_SWEEP=0
for d in 0 1 2 3; do
  LZ77X_WALK_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wd$d -o w -- python tests/gpu_time.py "$@" > gpurun_out/wd$d.log 2>&1
  python - $d <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/wd%s/w_kernel_stats.csv" % sys.argv[1])):
    if r["Name"].startswith(("k_walk", "void k_match")):
        print("debug", sys.argv[1], r["Name"][:24], "calls", r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
PY
  rm -rf gpurun_out/wd$d
done
