#!/bin/bash
# Round-4 rollout A/Bs on ONE box (interleaved): the re-tiled trunk / head kernels and the in-place
# action hand-off against round 3's device chain, then worker / group counts on the new chain.
# usage: scripts/r4_rollout_ab.sh <out.jsonl> [quick]
OUT=${1:-gpurun_out/r4_rollout_ab.jsonl}
: > $OUT
source "$(dirname "$0")/r4_lib.sh"
run v2_default
RLPYT_ROLLOUT_V1=1 run v1_kernels
run v2_no_zero_copy --no-zero-copy
RLPYT_ROLLOUT_V1=1 run r3_chain --no-zero-copy
run v2_default_again
if [ "$2" != "quick" ]; then
run workers16 --workers 16
run workers24 --workers 24
run workers32 --workers 32
run groups2 --groups 2
run groups3 --groups 3
run groups6 --groups 6
run groups8_w32 --groups 8 --workers 32
fi
cat $OUT
