# A/B of the fused launch train (KBA_UNFUSED_TRAIN=1: the round-5 sequence of launches), alternating on one box
for v in 1 0 1 0; do
  KBA_NO_SCHUR_PAIR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > /tmp/b_$v.json 2>/tmp/b_$v.err || tail -3 /tmp/b_$v.err
  python - <<PY
import json
d=json.load(open("/tmp/b_$v.json"))
bs=d.get("batch_sizes",{})
print("NO_PAIR=$v  %7.0f windows/s %7.2f ms/step | B=1: %.1f B=64: %.0f B=1024: %.0f | single %.2f ms | conv %d iters %.2f" % (d["value"], d["ms_per_step"], bs.get("1",{}).get("value",0), bs.get("64",{}).get("value",0), bs.get("1024",{}).get("value",0), d.get("single_window",{}).get("ms_per_solve_median",0), d["config"]["converged"], d["config"]["mean_lm_iterations"]))
PY
done
