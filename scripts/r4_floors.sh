#!/bin/bash
# Where a rollout time step goes: the default run beside (a) no device work in the step
# (RLPYT_NULL_STEP=1: host floor), (b) a frozen env (no env cost: framework + device floor),
# (c) both (the hand-off floor).  usage: scripts/r4_floors.sh <out.jsonl>
OUT=${1:-gpurun_out/r4_floors.jsonl}
: > $OUT
source "$(dirname "$0")/r4_lib.sh"
run default
RLPYT_NULL_STEP=1 run null_step
run frozen_env --frozen-env
RLPYT_NULL_STEP=1 run null_step_frozen_env --frozen-env
run frozen_env_w8 --frozen-env --workers 8
run frozen_env_g2 --frozen-env --groups 2
run frozen_env_g1 --frozen-env --groups 1
RLPYT_NULL_STEP=1 run null_step_w32 --workers 32
RLPYT_NULL_STEP=1 run null_step_w48 --workers 48
RLPYT_ROLLOUT_V1=1 run frozen_env_v1 --frozen-env
run default_again
cat $OUT
