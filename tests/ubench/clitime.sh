# end-to-end CLI timing (process start + file I/O + PCIe + GPU): bash tests/ubench/clitime.sh
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'.')
from lz77_amd import synth
synth.text(100_000_000, 0x5EED0001).tofile('/dev/shm/s1.bin')
PY
t() { local s=$(date +%s.%N); "$@"; local e=$(date +%s.%N); echo "$(echo "$e - $s" | awk '{print $1-$3}') s : $*" | cut -c1-60; }
for i in 1 2 3; do t ./lz77_amd/lz77 -c -i /dev/shm/s1.bin -o /dev/shm/s1.lz; done
for i in 1 2 3; do t ./lz77_amd/lz77 -d -i /dev/shm/s1.lz -o /dev/shm/s1.out; done
cmp /dev/shm/s1.bin /dev/shm/s1.out && echo RT_OK; ls -la /dev/shm/s1.lz; rm -f /dev/shm/s1.*
