# A/B of two library builds alternating on one box: scripts/gpu_ab_libs.sh TAG1:LIB1 TAG2:LIB2   (LIB empty = the product build)
for rep in 1 2; do for cfg in "$@"; do
  tag=${cfg%%:*}; lib=${cfg#*:}
  env ${lib:+LIMO_HIP_LIB=$PWD/$lib} python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > /tmp/b_$tag.json 2>/tmp/b_$tag.err || tail -3 /tmp/b_$tag.err
  python - <<PY
import json
d=json.load(open("/tmp/b_$tag.json"))
bs=d.get("batch_sizes",{})
print("%-10s %7.0f windows/s %7.2f ms/step | B=64: %.0f B=1024: %.0f | single %.2f ms | conv %d iters %.2f" % ("$tag", d["value"], d["ms_per_step"], bs.get("64",{}).get("value",0), bs.get("1024",{}).get("value",0), d.get("single_window",{}).get("ms_per_solve_median",0), d["config"]["converged"], d["config"]["mean_lm_iterations"]))
PY
done; done
