for rep in 1 2 3; do
for G in 0 3 4 5 6; do
  echo -n "rep $rep group_shift $G: "; PBRT_HIP_PATH_GROUP=$G timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print(l['value'], 'Mrays/s', l['ms_per_step'], 'ms/step trace', round(l['roofline']['avg_launch_ms']*6,1), 'film_check', l['config']['film_check']['ok'])"
done
done
