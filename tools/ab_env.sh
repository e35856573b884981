#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh <tag> "<ENV=VALUE ...>" [rounds]  -- same-box A/B of bench config 1 (tape route) with and
# without the given environment switches, alternating
tag=$1; envs=$2; rounds=${3:-2}
for r in $(seq $rounds); do
  timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe > gpurun_out/${tag}_base_$r.json 2> gpurun_out/${tag}_base_$r.err || exit 1
  echo "default $r: $(python tools/benchsum.py gpurun_out/${tag}_base_$r.json | head -1)"
  env $envs timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe > gpurun_out/${tag}_alt_$r.json 2> gpurun_out/${tag}_alt_$r.err || exit 1
  echo "$envs $r: $(python tools/benchsum.py gpurun_out/${tag}_alt_$r.json | head -1)"
done
