#!/bin/bash
# the whole GPU suite in one process under LZ77X_POISON=1 (every cached / fresh buffer 0xA5 at every lease), then a long fuzz under poison
mkdir -p gpurun_out
LZ77X_POISON=1 bash tools/suite_one_process.sh r06_suite_poison | tail -8
echo "== fuzz under poison"; LZ77X_POISON=1 timeout 800 python tests/gpu_fuzz_long.py 700 11000 2>&1 | tail -2
