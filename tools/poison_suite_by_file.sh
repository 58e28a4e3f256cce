#!/bin/bash
# LZ77X_POISON=1 per test file, failures listed (no -x); a file whose process dies is reported with its last test
mkdir -p gpurun_out
export LZ77X_POISON=1 PYTHONFAULTHANDLER=1
for f in ${FILES:-tests/test_gpu_decode_ranges.py tests/test_gpu_fuzz.py tests/test_gpu_memory.py tests/test_gpu_full.py tests/test_gpu_parity.py tests/test_gpu_soak.py}; do
  b=$(basename $f .py)
  timeout 1500 python -m pytest $f -m gpu -q --capture=sys -p no:cacheprovider --tb=line -v ${PYTEST_ARGS} > gpurun_out/r06_pl_$b.txt 2> gpurun_out/r06_pl_$b.err
  echo "== $f rc=$?"
  grep -E "FAILED|ERROR|passed|failed" gpurun_out/r06_pl_$b.txt | tail -40
  grep -E "Memory access fault|terminate|free\(\)|malloc\(\)|corrupt" gpurun_out/r06_pl_$b.err | head -5
  tail -3 gpurun_out/r06_pl_$b.txt | cut -c1-300
done
