"""k_lin_lm under ablation: the full-batch launch time of every library variant built by scripts/build_ablations.sh
(-DKBA_ABLATE=<n>: pieces of the kernel left out, results wrong by design) next to the product build.
usage (GPU box): python scripts/gpu_lin_ablate.py [B]   - child mode: python scripts/gpu_lin_ablate.py --child PICKLE TAG"""
import glob, os, pickle, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from limo_amd import ba, default_options
    ws = pickle.load(open(sys.argv[2], "rb"))
    ctx = ba.Context(0)
    opts = default_options(max_num_iterations=0, num_trim_rounds=0)
    b = ba.Batch(ctx, ws)
    for _ in range(3):
        b.reset(); b.solve(opts)
    b.kernel_stats(reset=True)
    N = 20
    for _ in range(N):
        b.reset(); b.solve(opts)
    st = b.kernel_stats()
    print("%-28s k_lin_lm %7.1f us per launch of %d windows" % (sys.argv[3], 1e3 * st["linearize_ms"] / N, len(ws)))
    sys.exit(0)

import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ws = bench.generate_windows([7000 + i for i in range(B)], 5, 2000)
pk = "/tmp/lin_ablate_windows.pkl"
pickle.dump(ws, open(pk, "wb"))
NAMES = {70: "steady-state body, nothing left out (no cost value, scale read)", 61: "no plane stores", 62: "no landmark tail", 63: "no cross-lane reduction", 64: "no pose Jacobian / U / g", 65: "planes as 2 x 16-byte stores",
         67: "no arithmetic at all (loads, stores, reduction, tail)", 68: "no landmark block (E, V, g)", 69: "61+62+63+64 (projection, residual, loss, E, V only)"}
variants = [("product", None)] + [("%d %s" % (int(os.path.basename(f)[12:-3]), NAMES.get(int(os.path.basename(f)[12:-3]), "")), f) for f in sorted(glob.glob(os.path.join(ROOT, "limo_amd/lib/ablate/liblimo_hip_*.so")))]
lib70 = os.path.join(ROOT, "limo_amd/lib/ablate/liblimo_hip_70.so")
extra = [("70 at 4 waves / SIMD (128 registers)", lib70, {"KBA_LIN_WAVES": "4"}), ("70 at 2 waves / SIMD (LDS pad)", lib70, {"KBA_LIN_LDS_PAD": "80000"})] if os.path.exists(lib70) else []
for tag, lib, more in [(t, l, {}) for t, l in variants] + extra + [("product (again)", None, {})]:
    env = dict(os.environ, KBA_GROUPS="1", **more)  # (kernel timing by events needs one slot group)
    if lib:
        env["LIMO_HIP_LIB"] = lib
    subprocess.call([sys.executable, os.path.abspath(__file__), "--child", pk, tag], env=env)
