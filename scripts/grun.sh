#!/bin/bash
# local helper: rebuild the gfx950 library (hipcc cross-compiles here), then run a script on the MI355X box
set -e
cd /root/repo
python -m micronet_amd.build | tail -1
bash tests/emu/build_emu.sh > /dev/null 2>&1 || true
timeout ${2:-1500} /usr/local/graft/bin/gpurun --timeout ${3:-1200} -- "bash $1" 2>&1 | tail -${4:-80}
