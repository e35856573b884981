#!/bin/bash
# usage: tools/prof_kernels.sh <tag>   (on the GPU box, from the repo root): rocprofv3 kernel statistics of the isolated
# per-shape launches that bench.py's roofline figures are measured on
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --kernels-only --profile-iters 100 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
echo "exit $?"
