#!/bin/bash
# Device-driven stepping (fetch kernel + enqueue-ahead serve loop) against the host-driven loop,
# interleaved on one box.  usage: scripts/r4_ahead.sh <out.jsonl>
OUT=${1:-gpurun_out/r4_ahead.jsonl}
: > $OUT
source "$(dirname "$0")/r4_lib.sh"
run ahead
RLPYT_DEVICE_FETCH=0 run host_driven
RLPYT_SERVE_AHEAD=0 run fetch_only
run ahead_again
run ahead_w16 --workers 16
run ahead_w24 --workers 24
run ahead_w32 --workers 32
run ahead_g2 --groups 2
run ahead_g3 --groups 3
run ahead_g6 --groups 6
run ahead_g8 --groups 8
run ahead_frozen --frozen-env
RLPYT_NULL_STEP=1 run ahead_null
RLPYT_DEVICE_FETCH=0 run host_driven_again
cat $OUT
