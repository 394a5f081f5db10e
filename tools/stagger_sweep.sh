#!/bin/bash
# Sweep of the second workgroup's start delay (TOUCAN_RB_STAGGER, units of 1 024 cycles) on the fused residual step, C = 64 and 32
# (two workgroups per CU).  Run on the MI355X box:  bash tools/stagger_sweep.sh > gpurun_out/stagger_sweep.txt
for s in ${STAGGERS:-0 4 8 12 16 20 24 32 48}; do
  echo "== TOUCAN_RB_STAGGER=$s"
  TOUCAN_RB_STAGGER=$s python tools/microbench_resblock.py --store bf16 --channels ${CHANNELS:-64,32} --acts ${ACTS:-snake,lrelu} --reps 10 || exit 1
done
