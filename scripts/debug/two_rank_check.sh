#!/bin/bash
# On the GPU box: 2 ranks on the one GPU over gloo with --check-params, N repetitions.
N=${1:-3}
for i in $(seq $N); do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --same-gpu --backend gloo --check-params --steps 2 --warmup 1 --no-kernel-timing --env-cost-leg-us 0 > /tmp/o.json 2> /tmp/o.err
  grep -o "check-params: all ranks.*\|ranks diverged.*\|non-finite.*" /tmp/o.err | sort -u | cut -c1-300
done
