#!/bin/bash
# rocprofv3 counter passes over the layout kernel on C4 at two run lengths (why are long runs slower?).  Run via gpurun.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $ROOT
for K in 8 64; do
  OUT=$ROOT/gpurun_out/prof/ndk$K
  mkdir -p $OUT
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 scripts/nd_pmc.py 2 0 $K > $OUT/run.log 2> $OUT/pmc_write.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 scripts/nd_pmc.py 2 0 $K > /dev/null 2> $OUT/pmc_fetch.err
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 scripts/nd_pmc.py 2 0 $K > /dev/null 2> $OUT/pmc_sq.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 scripts/nd_pmc.py 2 0 $K > /dev/null 2> $OUT/pmc_tcc.err
  cat $OUT/run.log
done
