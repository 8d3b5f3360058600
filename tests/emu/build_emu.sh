#!/bin/bash
# Build the CPU SIMT-emulation of libmicronet (TEST TOOL; see tests/emu/include/hip/hip_runtime.h)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tests/emu/build
CXX="g++ -O2 -std=c++17 -fPIC -ffp-contract=off -I tests/emu/include -x c++"
SRCS="quant_kernels conv_kernels qgemm_kernels qgemm_kxk qgemm_sign qgemm_pwb qgemm_k3s qgemm_dense conv_first optim_kernels norm_kernels iao_ops qact_kernels data_kernels linear_kernels iao_bnfuse iao_g3 iao_thin"
pids=""
for f in $SRCS; do
  extra=""; [ $f = quant_kernels ] && extra="-DMN_EMU_MAIN"
  $CXX $extra -c micronet_amd/csrc/$f.hip -o tests/emu/build/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done          # a failed compile fails the build (set -e), never links a stale object
objs=""; for f in $SRCS; do objs="$objs tests/emu/build/$f.o"; done
g++ -shared -o tests/emu/build/libmicronet_emu.so $objs
echo built tests/emu/build/libmicronet_emu.so
