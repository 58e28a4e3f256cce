#!/bin/bash
# usage: bash tools/prof_cmd.sh <tag> [ENV=VAL ...] -- <command ...>
# per-kernel durations (rocprofv3 --kernel-trace --stats) of any command; the summary lands in gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pc_$$; rm -rf $out; mkdir -p $out
env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- "$@" > gpurun_out/${tag}.log 2>&1
f=$(find $out -name t_kernel_stats.csv | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python - gpurun_out/${tag}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-70s calls %6s total_ms %9.3f avg_us %10.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
rm -rf $out
