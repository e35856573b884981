#!/bin/bash
# usage: tools/pmc.sh <tag> <kind> <N> <hs> [opts]   -- two PMC passes for one kernel shape
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_a -- python3 $GRAFT_REPO_ROOT/tools/onekernel.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_b -- python3 $GRAFT_REPO_ROOT/tools/onekernel.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_b.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_c -- python3 $GRAFT_REPO_ROOT/tools/onekernel.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_c.log 2>&1 || exit 1
echo ok
