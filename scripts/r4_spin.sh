#!/bin/bash
# Hand-off latency experiments: how long the env workers poll for the master's actions before they
# sleep in the futex (RLPYT_WORKER_SPIN iterations of ~10-40 ns), per worker count.
# usage: scripts/r4_spin.sh <out.jsonl>
OUT=${1:-gpurun_out/r4_spin.jsonl}
: > $OUT
source "$(dirname "$0")/r4_lib.sh"
run default
RLPYT_WORKER_SPIN=3000 run spin3k
RLPYT_WORKER_SPIN=30000 run spin30k
RLPYT_WORKER_SPIN=300000 run spin300k
RLPYT_WORKER_SPIN=30000 run spin30k_w16 --workers 16
RLPYT_WORKER_SPIN=30000 run spin30k_w14 --workers 14
RLPYT_WORKER_SPIN=30000 run spin30k_w24 --workers 24
RLPYT_WORKER_SPIN=30000 run spin30k_w32 --workers 32
RLPYT_WORKER_SPIN=30000 run spin30k_w16_g3 --workers 16 --groups 3
RLPYT_WORKER_SPIN=30000 run spin30k_w20_g6 --workers 20 --groups 6
RLPYT_WORKER_SPIN=30000 run spin30k_frozen --frozen-env
RLPYT_WORKER_SPIN=30000 RLPYT_NULL_STEP=1 run spin30k_null
run default_again
cat $OUT
