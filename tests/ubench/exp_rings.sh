# huge-page / ring sweep for the bench workload: bash tests/ubench/exp_rings.sh
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --no-cpu-baseline --steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['encode_breakdown_ms']['total_ms'], d['encode_breakdown_ms']['host_stageb_ms'], d['encode_breakdown_ms']['copy_ms'])"; }
run thp; run thp; LZ77X_HUGEPAGES=0 run nothp; LZ77X_HUGEPAGES=0 run nothp
bash tests/ubench/clitrace.sh 2>&1 | tail -6
