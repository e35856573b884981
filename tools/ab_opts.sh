#!/bin/bash
# usage (GPU box, repo root): tools/ab_opts.sh <tag> "<--opt a=b ...>" ["<--opt ...>" ...]  -- bench config 1 (tape route) with each
# set of kernel-variant overrides and without any, interleaved, two rounds
tag=$1; shift
for r in 1 2; do
  timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe --profile-iters 3 > gpurun_out/${tag}_base_$r.json 2> gpurun_out/${tag}_base_$r.err || exit 1
  echo "default $r: $(python tools/benchsum.py gpurun_out/${tag}_base_$r.json 2>/dev/null | head -1)"
  i=0
  for opts in "$@"; do
    i=$((i+1))
    timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe --profile-iters 3 $opts > gpurun_out/${tag}_alt${i}_$r.json 2> gpurun_out/${tag}_alt${i}_$r.err || { echo "$opts failed"; tail -3 gpurun_out/${tag}_alt${i}_$r.err; continue; }
    echo "$opts $r: $(python tools/benchsum.py gpurun_out/${tag}_alt${i}_$r.json 2>/dev/null | head -1)"
  done
done
