#!/bin/bash
# The whole GPU suite in ONE process, as the driver runs it, with the process's native stderr kept (--capture=sys leaves fd 2
# alone: a GPU memory fault, a glibc heap diagnostic or a std::terminate message lands in the file instead of in pytest's capture).
#   gpurun --timeout 2400 -- bash tools/suite_one_process.sh [tag]
tag=${1:-suite}
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout 2200 python -m pytest tests -m gpu -q --capture=sys -p no:cacheprovider --durations=12 > gpurun_out/${tag}.txt 2> gpurun_out/${tag}_stderr.txt
echo "rc=$?" >> gpurun_out/${tag}.txt
tail -30 gpurun_out/${tag}.txt
echo ---- stderr
grep -v "amdgpu.ids" gpurun_out/${tag}_stderr.txt | head -20
