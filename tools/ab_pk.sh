#!/bin/bash
# A/B on one box: bench config 1 with the shipped library (no packed-fp32 instructions) and with tools/libpgan_hip_pk.so
# (same sources built with packed-fp32 instructions allowed -- PGAN_PACKED_FP32=1 python __graft_entry__.py --force, the
# result copied there, then the default build restored; not tracked), alternating twice.
L=pg-alignments-gan_amd/libpgan_hip.so
cp $L /tmp/lib_nopk.so
for round in 1 2; do
  for v in nopk pk; do
    if [ $v = pk ]; then cp tools/libpgan_hip_pk.so $L; else cp /tmp/lib_nopk.so $L; fi
    timeout -k 10 200 python bench.py --config 1 --route tape --no-cpu-baseline --no-fp32-pipe > gpurun_out/s11_ab_${v}_$round.json 2> gpurun_out/s11_ab_${v}_$round.err || exit 1
    echo "$v $round: $(python tools/benchsum.py gpurun_out/s11_ab_${v}_$round.json | head -1)"
  done
done
cp /tmp/lib_nopk.so $L
