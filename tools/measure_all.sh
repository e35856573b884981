#!/bin/bash
# One box: the bench lines of configs 1-4 (+ config 2 at batch 256), then the two rocprofv3 kernel-statistics runs.
set -o pipefail
for c in 1 2 3 4; do
  timeout -k 10 400 python bench.py --config $c > gpurun_out/r02f_config$c.json 2> gpurun_out/r02f_config$c.err || { echo "config $c failed"; tail -5 gpurun_out/r02f_config$c.err; exit 1; }
  echo "config $c: $(python tools/benchsum.py gpurun_out/r02f_config$c.json | head -1)"
done
timeout -k 10 400 python bench.py --config 2 --batch 256 > gpurun_out/r02f_config2_b256.json 2> gpurun_out/r02f_config2_b256.err || exit 1
echo "config 2 b256: $(python tools/benchsum.py gpurun_out/r02f_config2_b256.json | head -1)"
bash tools/prof.sh r02f_prof && bash tools/prof_kernels.sh r02f_profk
python tools/stats.py r02f_prof 12
