"""A few FULL rounds of the streaming solve (1024 C2 windows in 1024 slots, every window iterating) for counter passes:
`rocprofv3 --pmc <counters> --kernel-trace -- python scripts/pmc_round.py` (scripts/gpu_pmc_r02.sh)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KBA_GROUPS"] = "1"
os.environ["KBA_SLOTS"] = "1024"
from limo_amd import synth

B = 1024


def _make(seed):
    return synth.make_window(seed)


# host processes generate the windows (10 ms each in numpy), forked before any GPU state exists
import multiprocessing as mp

with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
    ws = pool.map(_make, [5000 + i for i in range(B)], chunksize=16)
from limo_amd import ba, default_options  # noqa: E402

ctx = ba.Context(0)
o = default_options(max_num_iterations=4, num_trim_rounds=0)
b = ba.Batch(ctx, ws)
for _ in range(2):
    b.reset()
    b.solve(o)
reps = b.download()
print(json.dumps({"windows": B, "observations": int(sum(w.n_obs for w in ws)), "landmarks": int(sum(w.n_lm for w in ws)),
                  "depth_observations": int(sum((w.obs_d > 0).sum() for w in ws)), "free_slots": 40}))
