#!/bin/bash
# where the S3 tie-break's time goes: kernel stats with / without the rank index, the variants' probe bits, FETCH_SIZE of the new kernel
mkdir -p gpurun_out
echo "== kernel stats, rank index"; bash tools/prof_cmd.sh r06_c2a ITERS=1 -- python tools/time_c2.py | grep -i "tokens\|k_hr\|k_sx\|k_xfer\|scan" 
for p in 1 2 4; do echo "== RANK_PROBE=$p (1 no deferred, 2 no buckets, 4 no long runs)"; LZ77X_RANK_PROBE=$p ITERS=2 python tools/time_c2.py 2>&1 | grep encode | tail -1 | cut -c1-200; done
echo "== PMC FETCH_SIZE"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/c2f; rm -rf $out; mkdir -p $out
ITERS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o p -- python tools/time_c2.py > $out/log 2>&1
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("gpurun_out/c2f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]; acc[k]+=float(r["Counter_Value"]); n[k]+=1
for k,v in sorted(acc.items(), key=lambda kv:-kv[1])[:14]:
    print("%-60s launches %4d  fetch (x2 corrected) %8.2f GB" % (k[:60], n[k], v*2*1024/1e9))
PY
rm -rf $out
