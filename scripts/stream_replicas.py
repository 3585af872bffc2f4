#!/usr/bin/env python3
"""BASELINE.json configs[4] on N GPUs: the streaming pipeline of ONE sequence is a serial chain (every frame's pose-only
adjustment reads what the previous solve wrote - mono_lidar.cpp:186-260), so a node runs N independent sequences, one
process and one GPU each ("replicas of sequences"; no collective):

    python scripts/stream_replicas.py --gpus 8 [--frames 4541] [--exe tests/cpp/_build/limo_stream_gpu]

Prints one JSON line: aggregate frames/s over the N sequences (frames of all sequences / wall time of the slowest), and
every sequence's own fps / ATE."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4541)
    ap.add_argument("--az", type=int, default=2000)
    ap.add_argument("--exe", default=os.path.join(ROOT, "tests", "cpp", "_build", "limo_stream_gpu"))
    args = ap.parse_args()
    procs = []
    t0 = time.perf_counter()
    for k in range(args.gpus):
        env = dict(os.environ, LIMO_DEVICE=str(k))
        cmd = [args.exe, "--frames", str(args.frames), "--az", str(args.az), "--seed", str(7 + k), "--quiet"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    seqs = []
    for k, p in enumerate(procs):
        out, err = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(err[-2000:])
            sys.exit("sequence %d failed (rc %d)" % (k, p.returncode))
        kv = {l.split()[0]: float(l.split()[1]) for l in out.splitlines() if len(l.split()) == 2 and l.split()[0] in ("frames", "fps", "ate_rmse", "ate_max", "depth_fraction")}
        seqs.append(kv)
    wall = time.perf_counter() - t0
    print(json.dumps({"metric": "streaming VO, independent sequences (one per GPU)", "n_gpus": args.gpus, "frames_per_sequence": args.frames,
                      "pipeline_frames_per_s_sum": sum(s["fps"] for s in seqs), "wall_s_incl_input_synthesis": wall, "sequences": seqs}))


if __name__ == "__main__":
    main()
