#!/bin/bash
# usage (GPU box): bash tools/ts_probe.sh -- where k_tokens_sorted's time and instructions go: the variants build leaves the
# kernel after a phase (LZ77X_TS_PROBE=1 set-up, 2 + phase A and scans, 3 + B1, 4 + B2, 0 all; wrong output for 1-4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for p in 1 2 3 4 0; do
  echo "== LZ77X_TS_PROBE=$p"
  bash tools/pmc_cmd.sh tsp$p "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "k_tokens_sorted" LZ77X_TS_PROBE=$p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-file-to-file --streams 1
  LZ77X_TS_PROBE=$p python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-file-to-file --streams 1 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k_tiebreak_ms', d['encode_breakdown_ms']['k_tiebreak_ms'])
except Exception:
    print('(no bench line: a probe leaves the kernel early, the round trip fails by design)')"
done
