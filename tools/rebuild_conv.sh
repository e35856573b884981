#!/bin/bash
# Recompile only the split-operand convolution source (both splits, in parallel) and relink the library.
set -e
cd "$(dirname "$0")/../pg-alignments-gan_amd/csrc"
# same flags as __graft_entry__.build (no packed-fp32 vector instructions in device code)
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops"
/opt/rocm/bin/hipcc $F -c conv_x3.hip -o conv_x3.o &
/opt/rocm/bin/hipcc $F -DPGAN_SPLIT_F16 -c conv_x3.hip -o conv_h2.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpgan_hip.so conv_mfma.o conv_x3.o loader.o norm.o sampling.o small.o tower1d.o conv_h2.o
