#!/bin/bash
# On the GPU box: 2 ranks on the one GPU over gloo with --check-params, under A/B environment toggles.
run() {
  local n=$1; shift
  for i in $(seq $n); do
    env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --same-gpu --backend gloo --check-params $EXTRA --no-kernel-timing --env-cost-leg-us 0 > /tmp/o.json 2> /tmp/o.err
    echo "== $* $EXTRA:"; grep -o "check-params: all ranks.*\|ranks diverged.*" /tmp/o.err | sort -u | cut -c1-900
  done
}
EXTRA="--steps 1 --warmup 0" run 2 RLPYT_SPLIT_GEMM=0 RLPYT_TRUNK_FUSION=0 RLPYT_CLIP_ADAM=0
EXTRA="--steps 2 --warmup 1" run 2 RLPYT_SPLIT_GEMM=0 RLPYT_TRUNK_FUSION=0 RLPYT_CLIP_ADAM=0
