#!/bin/bash
# usage (GPU box, repo root): tools/ab_envs.sh <tag> "<ENV=VALUE ...>" ["<ENV=VALUE ...>" ...]  -- bench config 1 (tape route) under each
# set of environment switches and without any, interleaved, two rounds
tag=$1; shift
for r in 1 2; do
  timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe > gpurun_out/${tag}_base_$r.json 2> gpurun_out/${tag}_base_$r.err || exit 1
  echo "default $r: $(python tools/benchsum.py gpurun_out/${tag}_base_$r.json 2>/dev/null | head -1)"
  i=0
  for envs in "$@"; do
    i=$((i+1))
    env $envs timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe > gpurun_out/${tag}_alt${i}_$r.json 2> gpurun_out/${tag}_alt${i}_$r.err || exit 1
    echo "$envs $r: $(python tools/benchsum.py gpurun_out/${tag}_alt${i}_$r.json 2>/dev/null | head -1)"
  done
done
