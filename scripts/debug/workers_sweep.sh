for w in ${WS:-16 20 25 16 20 25 16 20 25}; do
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --env-cost-leg-us 0 --workers $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('workers', $w, round(d['value']), round(d['sampler']['ms_per_time_step'],4))"
done
