#!/bin/bash
# usage: bash tools/prof_kernel.sh <kernel-name-regex> [ENV=VAL ...]: per-kernel durations of one bench step under rocprofv3
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pk_$$; rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --streams 1 > $out/log 2>&1
python - $out "$pat" <<'PY'
import csv,sys,glob,re
out,pat=sys.argv[1],sys.argv[2]
for f in glob.glob(out+"/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Name"]): print(r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"])/1e3, "min_us", float(r["MinNs"])/1e3)
PY
rm -rf $out
