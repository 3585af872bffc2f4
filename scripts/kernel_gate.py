"""Per-kernel regression gate: compares rocprofv3 kernel tables (scripts/prof_summary.py output) of a BASE build
and a NEW build taken on ONE box and fails when a kernel of the new build is more than --tol (default 8 %) slower.

    python scripts/kernel_gate.py --base base1.txt [base2.txt ...] --new new.txt [--tol 0.08] [--min-us 5]

Several base tables (the A/B/A pattern of scripts/gpu_prof_two.sh: base, new, base) give the box's own run-to-run spread: a kernel only
fails when it is slower than the SLOWEST base run by more than tol.  Kernels are matched by name with template arguments and parameter
lists stripped ("k_lin_lm<3, true>(...)" -> "k_lin_lm"); variants of one kernel are summed (per round: total_ms / rounds, rounds = the
call count of k_sched_scan), so a kernel that was split, merged or re-templated is compared by what a round spends in it.  --map
"old=new[+new2]" compares renamed / merged kernels (e.g. --map k_cam_assemble+k_cam_solve=k_cam).  Kernels below --min-us per round
are reported, not gated (launch-latency noise).  Exit code 1 on a regression.
"""
import argparse
import re
import sys


def load(path):
    rows, rounds = {}, None
    for line in open(path):
        m = re.match(r"^(.{64})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
        if not m:
            continue
        name = m.group(1).strip()
        name = re.sub(r"^void\s+", "", name)
        name = re.sub(r"^kba::", "", name)
        base = re.split(r"[<(]", name)[0].strip()
        calls, total_ms = int(m.group(2)), float(m.group(3))
        rows[base] = rows.get(base, 0.0) + total_ms
        if base == "k_sched_scan":
            rounds = calls
    if not rounds:
        raise SystemExit("%s: no k_sched_scan row (not a streaming-solve table)" % path)
    return {k: 1e3 * v / rounds for k, v in rows.items()}, rounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", nargs="+", required=True)
    ap.add_argument("--new", required=True)
    ap.add_argument("--tol", type=float, default=0.08)
    ap.add_argument("--min-us", type=float, default=5.0)
    ap.add_argument("--map", action="append", default=[])
    ap.add_argument("--report-only", default="k_trim_residual,k_trim_max,k_trim_select",
                    help="kernels that run on the side stream UNDER the main stream's kernels: their durations depend on what they overlap with - reported, not gated")
    a = ap.parse_args()
    bases = [load(p)[0] for p in a.base]
    new, _ = load(a.new)
    groups = []  # (label, base names, new names)
    used_b, used_n = set(), set()
    for m in a.map:
        old, nw = m.split("=")
        bn, nn = old.split("+"), nw.split("+")
        groups.append((m, bn, nn))
        used_b.update(bn)
        used_n.update(nn)
    names = sorted(set(new) | set().union(*[set(b) for b in bases]))
    for k in names:
        if k in used_b or k in used_n or k.startswith("__amd"):
            continue
        groups.append((k, [k], [k]))
    bad = []
    print("%-44s %12s %12s %8s" % ("kernel (us per round)", "base", "new", "ratio"))
    tb = tn = 0.0
    for label, bn, nn in groups:
        b_runs = [sum(b.get(k, 0.0) for k in bn) for b in bases]
        b_hi, b_lo = max(b_runs), min(b_runs)
        n = sum(new.get(k, 0.0) for k in nn)
        tb += sum(b_runs) / len(b_runs)
        tn += n
        if b_hi == 0.0 and n == 0.0:
            continue
        ratio = n / b_hi if b_hi > 0 else float("inf")
        flag = ""
        if b_hi > 0 and n > (1.0 + a.tol) * b_hi:
            if label in a.report_only.split(","):
                flag = "  (side stream: not gated)"
            elif max(n, b_hi) >= a.min_us:
                flag = "  <-- REGRESSION"
                bad.append(label)
            else:
                flag = "  (below --min-us: not gated)"
        elif b_hi == 0.0:
            flag = "  (new kernel)"
        print("%-44s %5.1f..%-6.1f %12.1f %8.3f%s" % (label, b_lo, b_hi, n, ratio, flag))
    print("%-44s %12.1f %12.1f %8.3f" % ("sum", tb, tn, tn / tb if tb else 0.0))
    if bad:
        print("kernel gate FAILED: %s" % ", ".join(bad))
        sys.exit(1)
    print("kernel gate passed (tolerance %.0f %%)" % (100 * a.tol))


if __name__ == "__main__":
    main()
