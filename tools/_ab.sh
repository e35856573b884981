timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "d1d or tower or position or engine" > gpurun_out/buf_tests.log 2>&1 || { tail -20 gpurun_out/buf_tests.log; exit 1; }
tail -2 gpurun_out/buf_tests.log
for r in 1 2 3; do
  PGAN_HIP_LIBRARY=tools/_pk/libpgan_head.so timeout -k 10 120 python -u tools/step_times.py > gpurun_out/ab_a_$r.log 2>&1 || exit 1
  timeout -k 10 120 python -u tools/step_times.py > gpurun_out/ab_b_$r.log 2>&1 || exit 1
done
