#!/bin/bash
# Build ablation variants of libmicronet_hip.so (k_pws_wgrad with parts compiled out) next to the product library.
# Usage: scripts/ablate_wgrad.sh 1 2 3 4 5 7   -> micronet_amd/lib/libmicronet_hip_dbg<N>.so
set -e
cd "$(dirname "$0")/.."
L=micronet_amd/lib
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -DMN_WG2_DBGC=$d -c micronet_amd/csrc/qgemm_sign.hip -o $L/qgemm_sign_dbg$d.o &
done
wait
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmicronet_hip_dbg$d.so $L/quant_kernels.o $L/conv_kernels.o $L/qgemm_kernels.o $L/qgemm_kxk.o $L/qgemm_sign_dbg$d.o $L/conv_first.o $L/optim_kernels.o $L/norm_kernels.o
done
ls -la $L/*.so
