#!/bin/bash
# PMC counters of ONE S3 encode (212 MB mixed, s=65535 l=255; tools/time_c2.py), one rocprofv3 pass per counter set, --kernel-trace only
# (MI355X_MICROARCH.md, HBM / rocprofv3):   gpurun --timeout 900 -- 'bash tools/c2_pmc.sh'   -> gpurun_out/<tag>c2/<tag>c2_bench_pmc_summary.csv
tag=${TAG:-r06}c2
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $pass | cut -d' ' -f1)
  ITERS=1 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -o p -- python tools/time_c2.py > $out/pmc_$name.log 2>&1
done
python tools/pmc_summary.py $out $tag > $out/pmc_summary.log 2>&1
rm -rf $out/pmc_*/
ls -la $out; head -5 $out/${tag}_bench_pmc_summary.csv | cut -c1-300
