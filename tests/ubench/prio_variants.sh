# host recurrence variants on the GPU box's CPU: bash tests/ubench/prio_variants.sh
cd $GRAFT_REPO_ROOT
python tests/ubench/mkps.py
gcc -O3 -march=native tests/ubench/prio_variants.c -o /tmp/prio_variants.bin && /tmp/prio_variants.bin /dev/shm/ps.bin 100000000
taskset -c 8 /tmp/prio_variants.bin /dev/shm/ps.bin 100000000 | tail -1
rm -f /dev/shm/ps.bin
