#!/bin/bash
# usage (GPU box): bash tools/bench_quick.sh [tag] [ENV=VAL ...] -- one short bench run, the numbers that matter on one line
tag=${1:-q}; shift
env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-file-to-file --streams 1 > gpurun_out/bq_$tag.log 2>&1
python - gpurun_out/bq_$tag.log <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value",d["value"],"sha",d["stream_sha_ok"],"rt",d["roundtrip_ok"],"iters",d.get("prio_iters"))
    print(d["encode_breakdown_ms"]); print(d["decode_breakdown_ms"])
except Exception as e:
    print("FAILED",e); print(open(sys.argv[1]).read()[-3000:])
PY
