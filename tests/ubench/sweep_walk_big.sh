# large-window walker run-length sweep (S3, C2, serial streams): bash tests/ubench/sweep_walk_big.sh [runs...]
for r in ${@:-512 1024 2048 4096}; do echo run $r; LZ77X_SERIAL=1 LZ77X_WALK_RUN_BIG=$r LZ77X_ITERS=2 LZ77X_SWEEP=0 timeout 300 python tests/gpu_time.py mixed 212000000 65535 255 2>&1 | grep -E "  enc" | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip()[4:])
print({k:round(d[k],2) for k in ('total_ms','k_match_ms','k_sort_ms','k_walk_ms')})"; done
