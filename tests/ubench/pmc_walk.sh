# SQ counters of the match-stage kernels (one encode, serial streams): bash tests/ubench/pmc_walk.sh [gpu_time.py args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export LZ77X_SERIAL=1 LZ77X_ITERS=1 LZ77X_SWEEP=0
ARGS="$@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_walk1 -o p1 -- python tests/gpu_time.py $ARGS > gpurun_out/pmc_walk1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_walk2 -o p2 -- python tests/gpu_time.py $ARGS > gpurun_out/pmc_walk2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_walk1", "gpurun_out/pmc_walk2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:30]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); cnt[k] += 1
        for k in acc:
            if k.startswith(("k_walk", "void k_match", "k_tokens", "void k_tokens")):
                print(k, "launches", cnt[k], {c: round(v) for c, v in acc[k].items()})
PY
find gpurun_out/pmc_walk1 gpurun_out/pmc_walk2 -type f ! -name "*.csv" -delete
find gpurun_out/pmc_walk1 gpurun_out/pmc_walk2 -name "*kernel_trace.csv" -delete
