cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for seg in 0 262144; do
LZ77X_DECODE_SEGMENT=$seg rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec$seg -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --streams 1 > /dev/null 2>&1
echo seg $seg
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_dec$seg/t_kernel_stats.csv")):
    if "dec" in r["Name"] or "k_scan" in r["Name"]: print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e6,3), round(float(r["TotalDurationNs"])/4e6,3))
PY
done
