#!/bin/bash
# Wave-cycle breakdown (SQ counters) of the three large kernels on configs B and C, and the throughput against
# batch size, into gpurun_out/wave_cycles.txt and gpurun_out/batch_size_curve.txt.
set -o pipefail
{ for cfg in B C; do
    echo "config $cfg (1e9 bytes): per wavefront, cycles (SQ_WAVE_CYCLES = resident, SQ_BUSY_CYCLES = the SQ had work,"
    echo "SQ_WAIT_ANY / SQ_WAIT_INST_ANY = waiting for anything / for an instruction to issue, SQ_ACTIVE_INST_VALU / _LDS = executing)"
    bash tools/pmc_busy.sh --config $cfg || exit 1
  done; } > gpurun_out/wave_cycles.txt
echo "wave cycles done"
bash tools/batch_size_curve.sh > gpurun_out/batch_size_curve.txt || exit 1
cat gpurun_out/batch_size_curve.txt
