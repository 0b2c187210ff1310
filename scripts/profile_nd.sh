#!/bin/bash
# rocprofv3 passes over the nD layout kernel on C4 (run on the GPU box via gpurun).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/nd
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/nd_pmc.py 2 > $OUT/run.log 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 scripts/nd_pmc.py 2 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 scripts/nd_pmc.py 2 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 scripts/nd_pmc.py 2 > /dev/null 2> $OUT/pmc_tcc.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 scripts/nd_pmc.py 2 > /dev/null 2> $OUT/pmc_sq.err
cat $OUT/run.log
