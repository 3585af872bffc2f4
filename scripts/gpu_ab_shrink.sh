for v in 1 0 1 0; do
  KBA_NO_GRID_SHRINK=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > /tmp/b_$v.json 2>/tmp/b_$v.err || tail -3 /tmp/b_$v.err
  python - <<PY
import json
d=json.load(open("/tmp/b_$v.json"))
bs=d.get("batch_sizes",{})
print("NO_SHRINK=$v  %7.0f windows/s %7.2f ms/step | B=1: %s B=64: %s B=1024: %s | single %.2f ms" % (d["value"], d["ms_per_step"], bs.get("1",{}).get("value"), bs.get("64",{}).get("value"), bs.get("1024",{}).get("value"), d.get("single_window",{}).get("ms_per_solve_median",0)))
PY
done
