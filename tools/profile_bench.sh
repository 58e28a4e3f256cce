#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh r01
# 1. python bench.py                                          -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats  -- bench.py            -> <tag>_bench_kernel_stats.csv
# 3. three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*), never combined with other trace
#    domains, per MI355X_MICROARCH.md                          -> <tag>_bench_pmc_summary.csv, traffic.json
# Everything is written to gpurun_out/profile_<tag>/ ; copy the summaries into profiles/ afterwards.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/profile_$tag
rm -rf $out && mkdir -p $out
python bench.py > $out/${tag}_bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --streams 1 > $out/trace.log 2>&1
cp $out/trace/t_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
grep -o '{"metric.*' $out/trace.log > $out/${tag}_bench_under_rocprof.json      # bench.py's own hipEvent timing in the profiled run
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --streams 1 > $out/pmc_$name.log 2>&1
done
python - $out $tag <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[5:] if k.startswith("void ") else k
        k = k.split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(k, f)].add(r["Dispatch_Id"])
nl = collections.defaultdict(int)
for (k, f), ids in launches.items():
    nl[k] = max(nl[k], len(ids))
cols = sorted({c for k in acc for c in acc[k]})
with open("%s/%s_bench_pmc_summary.csv" % (out, tag), "w") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "launches_per_encode_plus_decode"] + [c + "_per_launch" for c in cols])
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
        w.writerow([k, nl[k]] + [round(acc[k].get(c, 0) / max(nl[k], 1), 1) for c in cols])
def hbm(k):
    a = acc.get(k)
    if not a: return None
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies 128-B read requests as 64 B -> fetch doubled
    return int((2 * a.get("FETCH_SIZE", 0) + a.get("WRITE_SIZE", 0)) * 1024 / max(nl[k], 1))
tr = {"round": tag,
      "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --streams 1",
      "correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM): fetch doubled; WRITE_SIZE as reported",
      "hbm_bytes_per_launch": {k.split("<")[0]: hbm(k) for k in sorted(acc) if k.startswith("k_")},
      "launches": {k.split("<")[0]: nl[k] for k in sorted(acc) if k.startswith("k_")}}
json.dump(tr, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(tr))
PY
rm -rf $out/trace $out/pmc_*/
head -c 600 $out/${tag}_bench.json; echo
head -8 $out/${tag}_bench_kernel_stats.csv | cut -c1-160
