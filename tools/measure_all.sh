#!/bin/bash
# One box: the bench lines of configs 1-4 (+ config 2 at batch 256), then the two rocprofv3 kernel-statistics runs and the
# per-shape reduction of the kernels-only trace.  usage: tools/measure_all.sh <prefix>   (files gpurun_out/<prefix>_*)
set -o pipefail
p=${1:-r03f}
for c in 1 2 3 4; do
  timeout -k 10 400 python bench.py --config $c > gpurun_out/${p}_config$c.json 2> gpurun_out/${p}_config$c.err || { echo "config $c failed"; tail -5 gpurun_out/${p}_config$c.err; exit 1; }
  echo "config $c: $(python tools/benchsum.py gpurun_out/${p}_config$c.json | head -1)"
done
timeout -k 10 400 python bench.py --config 2 --batch 256 > gpurun_out/${p}_config2_b256.json 2> gpurun_out/${p}_config2_b256.err || exit 1
echo "config 2 b256: $(python tools/benchsum.py gpurun_out/${p}_config2_b256.json | head -1)"
bash tools/prof.sh ${p}_prof && bash tools/prof_kernels.sh ${p}_profk || exit 1
python tools/stats.py ${p}_prof 12
python tools/trace_shapes.py ${p}_profk gpurun_out/${p}_profk.log > gpurun_out/${p}_trace_shapes.txt 2>&1; head -30 gpurun_out/${p}_trace_shapes.txt
