#!/bin/bash
# (ON THE GPU BOX) where the HOST time of a DQN / R2D1 bench iteration goes: cProfile over the whole bench
# process (fill iterations included -- they run the same sampler code; the update-only functions are the
# timed region's), sorted by own time and by cumulative time.
#   usage: scripts/debug/dqn_host_profile.sh [tag] [config=dqn] [extra bench args]
TAG=${1:-r6_hostprof}; CFG=${2:-dqn}; shift; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
case $CFG in
  dqn)  ARGS="--config dqn --replay-fill-itrs 100 --steps 2000 --warmup 20";;
  r2d1) ARGS="--config r2d1 --replay-fill-itrs 20 --steps 40 --warmup 3";;
esac
python - $ARGS --no-cpu-baseline "$@" > $OUT/bench_$CFG.json 2> $OUT/profile_$CFG.txt <<'PY'
import cProfile, pstats, sys, io
import bench
prof = cProfile.Profile()
prof.enable()
try:
    bench.replay_config_main(bench.parse())
finally:
    prof.disable()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(prof, stream=s).sort_stats(key).print_stats(70)
        sys.stderr.write(s.getvalue())
PY
tail -1 $OUT/bench_$CFG.json | cut -c1-300
head -100 $OUT/profile_$CFG.txt
