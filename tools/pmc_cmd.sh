#!/bin/bash
# usage: bash tools/pmc_cmd.sh <tag> "<counters>" <kernel-regex> [ENV=VAL ...] -- <command ...>
# one --pmc pass (kernel trace only, never with other trace domains); prints per-kernel averages per dispatch
tag=$1; ctrs=$2; pat=$3; shift 3
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pm_$$; rm -rf $out; mkdir -p $out
env "${envs[@]}" rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o p -- "$@" > gpurun_out/${tag}.log 2>&1
python - $out "$pat" <<'PY'
import csv,sys,glob,re,collections
out,pat=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
for f in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not re.search(pat,k): continue
        k=k.split("(")[0][:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k in acc:
    n=max(len(nd[k]),1)
    print(k, "dispatches", n, {c: round(v/n,1) for c,v in sorted(acc[k].items())})
PY
rm -rf $out
