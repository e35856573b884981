#!/bin/bash
# usage (GPU box, repo root): tools/optcmp.sh "<conv family prefix>" "--opt a=b --opt c=d" ...  -- per-shape timings of
# bench.py --kernels-only (engine launches: fp16 split, bounds given, graph-replayed) with each option set, side by side
fam=$1; shift
i=0
for o in "" "$@"; do
  timeout -k 10 200 python bench.py --kernels-only --profile-iters 50 $o > gpurun_out/optcmp_$i.json 2> gpurun_out/optcmp_$i.err || exit 1
  python - "$fam" "$o" gpurun_out/optcmp_$i.json <<'PY'
import json, sys
fam, o, f = sys.argv[1:4]
d = json.loads(open(f).read().strip().splitlines()[-1])
print(o or "default", {k.replace("conv_", ""): v["avg_us"] for k, v in d["roofline"]["per_shape"].items() if k.startswith(fam) and "Cl=1," not in k})
PY
  i=$((i+1))
done
