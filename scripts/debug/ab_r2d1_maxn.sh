OUT=$PWD/gpurun_out/r6o_ab_r2d1_maxn; mkdir -p $OUT; rm -f $OUT/ab.jsonl
for rep in 1 2; do for v in 1024 32768; do
RLPYT_DQN_CONVS_MAX_N=$v timeout 300 python bench.py --config r2d1 --replay-fill-itrs 60 --steps 15 --no-cpu-baseline 2> $OUT/r2d1_${v}_${rep}.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(cfg='r2d1', max_n=$v, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), updates_per_s=round(d.get('updates_per_s') or 0,2), sampling_frac=round(d.get('sampling_frac_of_step',0),3))))" | tee -a $OUT/ab.jsonl
done; done
