#!/bin/bash
# Instruction counts by phase of k_plan (configs B, C, D) and k_emit (B, C) into gpurun_out/phase_counts/.
#   bash tools/collect_phase_counts.sh
set -o pipefail
OUT=gpurun_out/phase_counts
mkdir -p $OUT
for cfg in B C D; do
  { echo "k_plan, config $cfg (1e9 bytes): dynamic instructions per 4 KiB wavefront, cumulative after each phase"
    echo "(tools/pmc_plan_phases.sh: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH with TRP_PROFILE_PLAN_STOP=n)."
    echo "1 staging, 2 classification, 3 block summaries + line carries, 4 plan_block, 5 in-tile gate 3, 6 plan stores, 0 whole kernel"
    bash tools/pmc_plan_phases.sh --config $cfg || exit 1; } > $OUT/config${cfg}_k_plan_instructions_by_phase.txt
  echo "k_plan $cfg done"
done
for cfg in B C; do
  { echo "k_emit, config $cfg (1e9 bytes): dynamic instructions per 2 KiB wavefront, cumulative after each phase"
    echo "(tools/pmc_emit_phases.sh with TRP_PROFILE_EMIT_STOP=n)."
    echo "1 loads + LDS staging, 2 prefix sums, 3 run copies into the output image, 6 token table staged in LDS, 4 token stores, 5 record offsets, 0 whole kernel"
    bash tools/pmc_emit_phases.sh --config $cfg || exit 1; } > $OUT/config${cfg}_k_emit_instructions_by_phase.txt
  echo "k_emit $cfg done"
done
