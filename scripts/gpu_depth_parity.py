"""GPU depth assignment vs oracle/depth_oracle.cpp, bit for bit, over many synthetic sweeps (run on the GPU box):
    python scripts/gpu_depth_parity.py [first_seed last_seed]
Prints one line per (seed, azimuth steps, ground labels) and a summary; exits non-zero on any differing bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import pyoracle  # noqa: E402
from limo_amd import ba, synth_lidar  # noqa: E402


def main():
    s0, s1 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 32)
    ctx = ba.Context(0)
    bad = 0
    total = 0
    for seed in range(s0, s1 + 1):
        for n_az in (2000, 4000):
            fr = synth_lidar.make_frame(seed)
            if n_az != 2000:
                fr["cloud"] = synth_lidar.make_sweep(seed, n_az=n_az)
                fr["uv"], fr["is_ground"], fr["z_true"] = synth_lidar.make_features(fr["cloud"], seed)
            for g in (False, True):
                dg = ba.depth_estimate(ctx, fr, use_ground_labels=g)
                do = pyoracle.depth_estimate(fr, use_ground_labels=g)
                diff = np.flatnonzero(dg.view(np.uint32) != do.view(np.uint32))
                msg = ""
                if g:
                    ng, pg = ba.depth_last_ground_plane(ctx, 0)
                    no, po = pyoracle.ground_plane(fr)
                    if ng != no or not np.array_equal(pg, po):
                        msg = "  PLANE DIFFERS gpu %d %r oracle %d %r" % (ng, pg, no, po)
                        bad += 1
                total += dg.size
                if diff.size:
                    bad += diff.size
                    msg += "  first diffs: " + ", ".join("#%d gpu %.9g oracle %.9g ground=%d" % (k, dg[k], do[k], fr["is_ground"][k]) for k in diff[:5])
                print("seed %2d n_az %d ground %d: %d features, %d with depth, %d differing%s" % (seed, n_az, g, dg.size, (do > 0).sum(), diff.size, msg), flush=True)
    print("SUMMARY: %d features compared, %d differing bits/planes" % (total, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
