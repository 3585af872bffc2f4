for pb in 192 384 768 192 384 768; do
  KBA_SCHUR_PAIR_BOUND=$pb python bench.py --steps 5 --warmup 2 --batch 1024 --no-cpu-baseline --no-extras --no-pmc > /tmp/g.json 2>/tmp/g.err || tail -3 /tmp/g.err
  python -c "
import json; d=json.load(open('/tmp/g.json')); print('B=1024 pair bound $pb  %7.0f windows/s %7.2f ms/step' % (d['value'], d['ms_per_step']))"
done
for pb in 192 768; do
  KBA_SCHUR_PAIR_BOUND=$pb python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > /tmp/g.json 2>/tmp/g.err || tail -3 /tmp/g.err
  python -c "
import json; d=json.load(open('/tmp/g.json')); print('B=16384 pair bound $pb  %7.0f windows/s %7.2f ms/step' % (d['value'], d['ms_per_step']))"
done
