#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_table.sh
# FETCH_SIZE and WRITE_SIZE of every convolution launch shape of one iteration: one rocprofv3 --pmc pass per counter and
# shape (tools/pmc_traffic.sh), collected into gpurun_out/pmc_table.json by tools/pmc_collect.py.
set -e
for spec in "down 64" "down 128" "up 64" "up 128" "up 320" "wgrad 64" "wgrad 128" "wgrad 256"; do
  set -- $spec
  for hs in 16 8 4; do
    bash tools/pmc_traffic.sh pmc_$1_$2_$hs $1 $2 $hs
    echo "$1 N=$2 hs=$hs done"
  done
done
python3 tools/pmc_collect.py > gpurun_out/pmc_table.json
echo collected
