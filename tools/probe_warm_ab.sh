for rep in 1 2 3; do
 for p in 40 150 400; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --probe-reps $p --cpu-seconds 0 --parity off 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('probe_reps $p', d['ms_per_step'], r['frac'], r['step_ms']['median'], r['step_ms']['first'], r['step_ms']['p90'])"
 done
done
