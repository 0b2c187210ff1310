#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box via gpurun).  Output under gpurun_out/prof/<tag>.
# usage: scripts/profile.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline $@"      # the driver runs: bench.py --gpus 1 --steps 20 --warmup 5
PARGS="$ARGS --no-extra-legs"                               # counter passes: the timed region only
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PARGS > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $PARGS > /dev/null 2> $OUT/pmc_write.err
echo "write done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 bench.py $PARGS > /dev/null 2> $OUT/pmc_tcc.err
echo "tcc done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PARGS > /dev/null 2> $OUT/pmc_sq.err
echo "sq done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $PARGS > /dev/null 2> $OUT/pmc_sq2.err || true
echo "sq2 done"
find $OUT -name "*.csv" | head -50
