#!/bin/bash
mkdir -p gpurun_out/r4j
for b in conv2_x.1 conv5_x.0; do
echo "==== $b"
timeout 300 python scripts/dbg_c5_whole.py $b 2>&1 | tail -40
echo "---- producer min/max hand-over off"
MN_NO_PRODUCER_MINMAX=1 timeout 300 python scripts/dbg_c5_whole.py $b 2>&1 | tail -40
done
