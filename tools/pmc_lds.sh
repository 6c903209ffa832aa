#!/bin/bash
# LDS counters of the bench kernels (optionally with TRP_PROFILE_EMIT_STOP set by the caller).
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lds_${1:-x}
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -f csv -d $OUT -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT.json 2> $OUT.log || exit 1
python3 tools/pmc_summary.py $OUT > $OUT.txt
grep -A9 "^k_emit<false>" $OUT.txt
