for b in 1024 2048 4096 8192; do for g in 2 3 2 3; do
  KBA_GROUPS=$g python bench.py --steps 5 --warmup 2 --batch $b --no-cpu-baseline --no-extras --no-pmc > /tmp/g.json 2>/tmp/g.err || tail -3 /tmp/g.err
  python -c "
import json; d=json.load(open('/tmp/g.json')); print('batch $b groups $g  %7.0f windows/s %7.2f ms/step' % (d['value'], d['ms_per_step']))"
done; done
