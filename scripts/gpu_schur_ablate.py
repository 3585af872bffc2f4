"""The Schur kernels under ablation (library variants of scripts/build_ablations.sh, -DKBA_ABLATE=8x): time of the Schur
launches of ONE full LM iteration over B windows (max_num_iterations = 1: linearise, one step; results wrong by design).
usage (GPU box): python scripts/gpu_schur_ablate.py [B]"""
import glob, os, pickle, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from limo_amd import ba, default_options
    ws = pickle.load(open(sys.argv[2], "rb"))
    ctx = ba.Context(0)
    opts = default_options(max_num_iterations=1, num_trim_rounds=0)
    b = ba.Batch(ctx, ws)
    for _ in range(3):
        b.reset(); b.solve(opts)
    b.kernel_stats(reset=True)
    N = 20
    for _ in range(N):
        b.reset(); b.solve(opts)
    st = b.kernel_stats()
    print("%-44s Schur kernels %7.1f us, k_lin_lm %7.1f us per solve of %d windows (one LM iteration)" % (sys.argv[3], 1e3 * st["schur_ms"] / N, 1e3 * st["linearize_ms"] / N, len(ws)))
    sys.exit(0)
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ws = bench.generate_windows([7000 + i for i in range(B)], 5, 2000)
pk = "/tmp/schur_ablate_windows.pkl"
pickle.dump(ws, open(pk, "wb"))
NAMES = {81: "no MFMA products", 82: "no fill arithmetic (M, xn/yn, Ft, F^T E Bt)", 83: "no landmark-side loads (c, p, Bt, g)"}
libs = sorted(glob.glob(os.path.join(ROOT, "limo_amd/lib/ablate/liblimo_hip_8*.so")))
for tag, lib in [("product", None)] + [("%s %s" % (os.path.basename(f)[12:-3], NAMES.get(int(os.path.basename(f)[12:-3]), "")), f) for f in libs] + [("product (again)", None)]:
    env = dict(os.environ, KBA_GROUPS="1")
    if lib:
        env["LIMO_HIP_LIB"] = lib
    subprocess.call([sys.executable, os.path.abspath(__file__), "--child", pk, tag], env=env)
