#!/bin/bash
# (ON THE GPU BOX) does the 16-CPU cgroup quota throttle the PPO bench?  cpu.stat (nr_throttled / throttled_usec)
# around short bench runs for several env-worker counts / worker spin settings.
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for cfg in "20 -" "16 -" "14 -" "12 -" "20 0" "16 0" "20 2000"; do
  set -- $cfg; W=$1; SPIN=$2
  if [ "$SPIN" = "-" ]; then unset RLPYT_WORKER_SPIN; else export RLPYT_WORKER_SPIN=$SPIN; fi
  for rep in 1 2; do
    A=$(stat)
    R=$(python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra-configs --env-cost-leg-us 0 --no-kernel-timing --workers $W 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sampler']
print(round(d['value']), round(d['ms_per_step'],2), round(s['ms_per_time_step'],4))")
    B=$(stat)
    python - "$A" "$B" "$W" "$SPIN" "$R" <<'PY'
import sys
def parse(s):
    t = s.split(); return {t[i]: int(t[i + 1]) for i in range(0, len(t) - 1, 2)}
a, b = parse(sys.argv[1]), parse(sys.argv[2])
d = {k: b[k] - a.get(k, 0) for k in b}
print(f"workers {sys.argv[3]} spin {sys.argv[4]}: sps/ms/time_step {sys.argv[5]} | usage {d.get('usage_usec',0)/1e6:.2f}s "
      f"periods {d.get('nr_periods',0)} throttled {d.get('nr_throttled',0)} throttled_s {d.get('throttled_usec',0)/1e6:.3f}")
PY
  done
done
