#!/bin/bash
# A/B on one box: bench config 1 with the shipped library and with tools/libpgan_hip_other.so (another build of the same
# C-ABI, e.g. an earlier kernel version; not tracked), alternating twice.
L=pg-alignments-gan_amd/libpgan_hip.so
cp $L /tmp/lib_new.so
for round in 1 2 3; do
  for v in new other; do
    if [ $v = other ]; then cp tools/libpgan_hip_other.so $L; else cp /tmp/lib_new.so $L; fi
    timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe > gpurun_out/ab_${v}_$round.json 2> gpurun_out/ab_${v}_$round.err || exit 1
    echo "$v $round: $(python tools/benchsum.py gpurun_out/ab_${v}_$round.json | head -1)"
  done
done
cp /tmp/lib_new.so $L
