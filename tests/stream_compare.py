"""Call-by-call comparison of two limo_stream drives (LIMO_STREAM_TRACE=1 lines of limo_amd/kba/stream_driver.hpp) - used to
hold the GPU drive against the drive with the ORACLE behind the C-ABI (tests/cpp/oracle_abi.cpp), SURVEY 8c / config 5.

A drive is a feedback loop: the poses of solve k decide which landmarks solve k + 1 selects.  While both drives select the same
SETS (the trace carries a hash of the selected ids) they optimise the same windows and the north-star bar applies call by call:
pose translation within 1e-4 relative.  After the first call whose selection differs the two drives solve different problems;
from there only the trajectory as a whole is compared."""
import numpy as np


def parse_trace(text):
    calls = []
    for l in text.splitlines():
        if not l.startswith("trace "):
            continue
        t = l.split()
        # trace <what> <stamp> cost <c> pose q0..q3 t0..t2 selected <n> set <hash>
        calls.append({"what": t[1], "stamp": int(t[2]), "cost": float(t[4]), "pose": np.array(t[6:13], float), "n_selected": int(t[14]),
                      "set": t[16] if len(t) > 16 else None})
    return calls


def vehicle_position(pose):
    """Translation of the inverse of (q, t): vehicle position in the origin frame."""
    q, t = pose[:4] / np.linalg.norm(pose[:4]), pose[4:]
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return -R.T @ t


def compare(a_text, b_text):
    """Returns dict: n_calls, first_divergence (index of the first call with another selection or None), its stamp / frame,
    max_rel_before (max over the calls before it of |dx| / max(path, 1 m)), max_abs_before, max_rel_cost_before."""
    a, b = parse_trace(a_text), parse_trace(b_text)
    assert len(a) == len(b) and len(a) > 0, (len(a), len(b))
    path, last, first_div = 0.0, None, None
    max_rel = max_abs = max_cost = 0.0
    for i, (x, y) in enumerate(zip(a, b)):
        assert x["what"] == y["what"] and x["stamp"] == y["stamp"], (i, x["what"], y["what"])
        pa, pb = vehicle_position(x["pose"]), vehicle_position(y["pose"])
        if last is not None:
            path += np.linalg.norm(pb - last) if x["what"] == "pose-only" else 0.0
        if x["what"] == "pose-only":
            last = pb
        if x["set"] != y["set"] or x["n_selected"] != y["n_selected"]:
            first_div = i
            break
        d = np.linalg.norm(pa - pb)
        max_abs = max(max_abs, d)
        max_rel = max(max_rel, d / max(path, 1.0))
        if y["cost"] > 0:
            max_cost = max(max_cost, abs(x["cost"] - y["cost"]) / max(abs(y["cost"]), 1e-300))
    out = {"n_calls": len(a), "first_divergence": first_div, "max_rel_before": max_rel, "max_abs_before": max_abs, "max_rel_cost_before": max_cost, "path_before": path}
    if first_div is not None:
        out["divergence_stamp"] = a[first_div]["stamp"]
        out["divergence_frame"] = int(a[first_div]["stamp"] // 50000000)
    return out
