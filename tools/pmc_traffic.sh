#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> <kind> <N> <hs>   -- FETCH_SIZE and WRITE_SIZE passes (separate runs) for one conv shape
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_f -- python3 $GRAFT_REPO_ROOT/tools/onekernel.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_f.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_w -- python3 $GRAFT_REPO_ROOT/tools/onekernel.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_w.log 2>&1 || exit 1
echo ok
