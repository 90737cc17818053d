#!/bin/bash
# A/B (ON THE GPU BOX) of the two-launch row split of the tall NT GEMM (the trunk's input gradient):
# RLPYT_GEMM_NT_R256 = -1 (one launch of 256-row tiles, as before round 6) / unset (gemm_nt_plan) / others
OUT=$PWD/gpurun_out/${1:-r6p_nt_split}; mkdir -p $OUT; rm -f $OUT/ab.jsonl
for rep in 1 2; do for r in -1 plan 28 27 24 18 0; do
  if [ $r = plan ]; then unset RLPYT_GEMM_NT_R256; else export RLPYT_GEMM_NT_R256=$r; fi
  python scripts/gemm_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(r256='$r', rep=$rep, dgrad_us=d['dgrad_nt_lockstep_on_transposed_w']['us'], fwd_us=d['fwd_nt_lockstep']['us'], wgrad_us=d['wgrad_tn_lockstep_split_k']['us'])))" | tee -a $OUT/ab.jsonl
done; done
