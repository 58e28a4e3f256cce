#!/bin/bash
# usage (GPU box): bash tools/gaps.sh [ENV=VAL ...] -- kernel timeline of ONE encode+decode step: where the stream sits idle between kernels
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/gp_$$; rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-file-to-file --streams 1 > $out/log 2>&1
python - $out <<'PY'
import csv,sys,glob
out=sys.argv[1]
ev=[]
for f in glob.glob(out+"/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]))
for f in glob.glob(out+"/**/t_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size",""))))
ev.sort()
# the last occurrence of k_c1_chunks starts the timed step's encode
starts=[i for i,e in enumerate(ev) if e[2].startswith("k_c1_chunks")]
i0=starts[-1]
# walk back to the pad kernel before it
while i0>0 and ev[i0][0]-ev[i0-1][1] < 200000: i0-=1
t0=ev[i0][0]; prev=ev[i0][0]; busy=0; gaps=0
for s,e,nm in ev[i0:]:
    gap=(s-prev)/1e3
    print(f"{(s-t0)/1e3:10.1f} us  +gap {gap:8.1f}  dur {(e-s)/1e3:9.1f}  {nm}")
    if gap>0: gaps+=gap
    busy+=(e-s)/1e3; prev=max(prev,e)
print("busy_us",busy,"gaps_us",gaps)
PY
rm -rf $out
