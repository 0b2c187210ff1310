#!/bin/bash
# rocprofv3 --kernel-trace --stats over the DEFAULT bench.py command (201 steps = one fused launch of 201 iterations, plus
# the untimed priming launch of the same shape).  Output under gpurun_out/prof/<tag>; condense with
#   python scripts/summarize_profile.py gpurun_out/prof/<tag> profiles/rNN/default --no-traffic
# usage: scripts/profile_default.sh <tag>
set -e
TAG=${1:-default}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
