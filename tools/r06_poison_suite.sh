#!/bin/bash
# The whole GPU suite with every cached buffer poisoned at the start of every call (LZ77X_POISON=1, ctx.cpp) and the native
# stderr of the process in a file (--capture=sys leaves fd 2 alone: a GPU memory fault, a glibc heap diagnostic or a
# std::terminate message is no longer swallowed with pytest's capture).  gpurun -- bash tools/r06_poison_suite.sh
mkdir -p gpurun_out
export LZ77X_POISON=1
export PYTHONFAULTHANDLER=1
timeout 2400 python -m pytest tests -m gpu -q --capture=sys -p no:cacheprovider --durations=15 \
    > gpurun_out/r06_poison_suite.txt 2> gpurun_out/r06_poison_stderr.txt
echo "rc=$?" >> gpurun_out/r06_poison_suite.txt
tail -60 gpurun_out/r06_poison_suite.txt
echo ---- stderr tail
tail -40 gpurun_out/r06_poison_stderr.txt
