#!/bin/bash
# Build the CPU SIMT-emulation of libmicronet (TEST TOOL; see tests/emu/include/hip/hip_runtime.h)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tests/emu/build
CXX="g++ -O2 -std=c++17 -fPIC -ffp-contract=off -I tests/emu/include -x c++"
$CXX -DMN_EMU_MAIN -c micronet_amd/csrc/quant_kernels.hip -o tests/emu/build/quant_kernels.o &
$CXX -c micronet_amd/csrc/conv_kernels.hip -o tests/emu/build/conv_kernels.o &
$CXX -c micronet_amd/csrc/qgemm_kernels.hip -o tests/emu/build/qgemm_kernels.o &
$CXX -c micronet_amd/csrc/qgemm_kxk.hip -o tests/emu/build/qgemm_kxk.o &
$CXX -c micronet_amd/csrc/qgemm_sign.hip -o tests/emu/build/qgemm_sign.o &
$CXX -c micronet_amd/csrc/conv_first.hip -o tests/emu/build/conv_first.o &
$CXX -c micronet_amd/csrc/optim_kernels.hip -o tests/emu/build/optim_kernels.o &
$CXX -c micronet_amd/csrc/norm_kernels.hip -o tests/emu/build/norm_kernels.o &
wait
g++ -shared -o tests/emu/build/libmicronet_emu.so tests/emu/build/quant_kernels.o tests/emu/build/conv_kernels.o tests/emu/build/qgemm_kernels.o tests/emu/build/qgemm_kxk.o tests/emu/build/qgemm_sign.o tests/emu/build/conv_first.o tests/emu/build/optim_kernels.o tests/emu/build/norm_kernels.o
echo built tests/emu/build/libmicronet_emu.so
