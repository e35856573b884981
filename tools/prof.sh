#!/bin/bash
# usage: tools/prof.sh <tag> [bench args]  (runs on the GPU box from the repo root)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-pipe --profile-iters 1 --steps 8 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
echo "exit $?"
