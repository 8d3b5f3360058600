#!/bin/bash
# local helper: rebuild the gfx950 library (hipcc cross-compiles here), then run scripts/gpu_check.sh <args> on the MI355X box
#   scripts/grun.sh "tests bench" [gpurun timeout s] [tail lines]
set -e
cd /root/repo
python -m micronet_amd.build | tail -1
bash tests/emu/build_emu.sh > /dev/null 2>&1 || true
T=${2:-1500}
timeout $((T + 900)) /usr/local/graft/bin/gpurun --timeout $T -- "bash scripts/gpu_check.sh $1" 2>&1 | tail -${3:-80}
