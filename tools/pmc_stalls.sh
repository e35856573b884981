#!/bin/bash
# usage (GPU box, repo root): tools/pmc_stalls.sh   -- wave-cycle breakdown (tools/pmc.sh: two SQ counter passes + one
# GRBM pass per shape, each its own rocprofv3 run) of the split-operand convolution kernels at the launch shapes that
# carry most of an iteration, summarised by tools/pmc_stalls.py into gpurun_out/pmc_stalls.txt
set -e
for spec in "up 128 4" "up 64 4" "up 128 16" "down 128 4" "down 64 4" "down 128 16" "wgrad 128 4" "wgrad 128 8" "wgrad 128 16"; do
  set -- $spec
  bash tools/pmc.sh st_$1_$2_$3 $1 $2 $3 > /dev/null
  echo "$spec done"
done
python3 tools/pmc_stalls.py > gpurun_out/pmc_stalls.txt
cat gpurun_out/pmc_stalls.txt
