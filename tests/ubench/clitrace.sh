cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'.')
from lz77_amd import synth
synth.text(100_000_000, 0x5EED0001).tofile('/dev/shm/s1.bin')
PY
s=$(date +%s.%N); LZ77X_TRACE=1 ./lz77_amd/lz77 -c -i /dev/shm/s1.bin -o /dev/shm/s1.lz; e=$(date +%s.%N); echo "total $(echo "$e $s" | awk '{print $1-$2}') s"
rm -f /dev/shm/s1.*
