# LDS walker run-length sweep (S1, C1): bash tests/ubench/sweep_walk.sh
for r in 256 512 768 1024 1536 2048; do echo run $r; LZ77X_WALK_RUN=$r LZ77X_ITERS=2 LZ77X_SWEEP=0 python tests/gpu_time.py 2>&1 | grep -E "  enc" | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip()[4:])
print({k:round(d[k],2) for k in ('k_match_ms','k_sort_ms','k_walk_ms','total_ms')})"; done
