set -e
P=pg-alignments-gan_amd
cp $P/libpgan_hip.so $P/libpgan_hip_new.so
for r in 1 2; do
  for w in new old; do
    cp $P/libpgan_hip_$w.so $P/libpgan_hip.so
    echo "== $w"
    timeout -k 10 150 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fp32-pipe 2>/dev/null > gpurun_out/tmp_$w.json; python tools/benchsum.py gpurun_out/tmp_$w.json | head -1
  done
done
cp $P/libpgan_hip_new.so $P/libpgan_hip.so
