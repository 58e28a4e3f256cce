#!/bin/bash
# usage (GPU box): bash tools/cli_trace.sh [bytes]  -- where the wall time of `lz77 -c` / `lz77 -d` goes (LZ77X_TRACE=1)
n=${1:-100000000}
d=/dev/shm/lz77x_trace; mkdir -p $d
python - $n $d/in <<'PY'
import sys
sys.path.insert(0, ".")
from lz77_amd import synth
synth.text(int(sys.argv[1]), 0x5EED0001).tofile(sys.argv[2])
PY
for rep in 1 2; do
  echo "== encode (run $rep)"; t0=$(date +%s%N); env LZ77X_TRACE=1 LZ77X_T0=$t0 lz77_amd/lz77 -c -i $d/in -o $d/z 2>&1 | grep -v "^$"; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
  echo "== decode (run $rep)"; t0=$(date +%s%N); env LZ77X_TRACE=1 LZ77X_T0=$t0 lz77_amd/lz77 -d -i $d/z -o $d/out 2>&1 | grep -v "^$"; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
done
cmp $d/in $d/out && echo roundtrip ok
rm -rf $d
