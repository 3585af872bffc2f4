"""ctypes binding of the test-only CPU emulation of the HIP pipeline (tests/cpp/emu_pipeline.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from limo_amd import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = [os.path.join(_HERE, "cpp", "emu_pipeline.cpp"), os.path.join(_HERE, "..", "limo_amd", "csrc", "kba_pack.cpp")]
LIB_PATH = os.path.join(_HERE, "cpp", "_build", "libkba_emu.so")
_lib = None


def build(force=False):
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    deps = _SRC + [os.path.join(_HERE, "..", "limo_amd", "csrc", f) for f in os.listdir(os.path.join(_HERE, "..", "limo_amd", "csrc")) if f.endswith(".hpp")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-fPIC", "-shared", "-o", LIB_PATH] + _SRC)


SHIM_SRC = [os.path.join(_HERE, "..", "limo_amd", "kba", "bundle_adjuster_keyframes.cpp")]
ABI_LIB_PATH = os.path.join(_HERE, "cpp", "_build", "libkba_emu_abi.so")
SHIM_TEST_EMU = os.path.join(_HERE, "cpp", "_build", "test_kba_shim_emu")
SHIM_TEST_GPU = os.path.join(_HERE, "cpp", "_build", "test_kba_shim_gpu")


ORACLE_ABI_LIB_PATH = os.path.join(_HERE, "cpp", "_build", "libkba_oracle_abi.so")


def build_oracle_abi():
    """tests/cpp/oracle_abi.cpp: the C-ABI names served by liboracle.so (the restated Ceres loop, NOT the emulated kernels) -
    the third backend of the shim tests and of limo_stream: a config-5 drive whose arithmetic shares nothing with the GPU's."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    oracle_dir = os.path.abspath(os.path.join(_HERE, "..", "oracle", "_build"))
    if not os.path.exists(os.path.join(oracle_dir, "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "..", "oracle")])
    src = os.path.join(_HERE, "cpp", "oracle_abi.cpp")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-o", ORACLE_ABI_LIB_PATH, src, "-L" + oracle_dir, "-loracle", "-Wl,-rpath," + oracle_dir])
    return ["-L" + oracle_dir, ORACLE_ABI_LIB_PATH, "-loracle", "-Wl,-rpath," + os.path.dirname(ORACLE_ABI_LIB_PATH), "-Wl,-rpath," + oracle_dir]


def build_stream_test(gpu=False, oracle=False):
    """tests/cpp/test_kba_stream.cpp (streaming sequence through the shim) against the emulated C-ABI, liblimo_hip.so, or the
    oracle behind the C-ABI."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    csrc = os.path.join(_HERE, "..", "limo_amd", "csrc")
    test_src = os.path.join(_HERE, "cpp", "test_kba_stream.cpp")
    out = os.path.join(_HERE, "cpp", "_build", "test_kba_stream_gpu" if gpu else "test_kba_stream_oracle" if oracle else "test_kba_stream_emu")
    if oracle:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", out, test_src] + SHIM_SRC + build_oracle_abi())
        return out
    if not gpu:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-fPIC", "-shared", "-DKBA_EMU_EXPORT_ABI", "-o", ABI_LIB_PATH] + _SRC + [os.path.join(csrc, "host_misc.cpp")])
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", out, test_src] + SHIM_SRC + [ABI_LIB_PATH, "-Wl,-rpath," + os.path.dirname(ABI_LIB_PATH)])
        return out
    libdir = os.path.join(_HERE, "..", "limo_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", out, test_src] + SHIM_SRC + ["-L" + libdir, "-llimo_hip", "-Wl,-rpath," + os.path.abspath(libdir)])
    return out


STREAM_APP_SRC = os.path.join(_HERE, "..", "apps", "limo_stream", "limo_stream.cpp")


def build_stream_app(gpu=False, oracle=False):
    """apps/limo_stream (synthetic drive through limo_amd/kba/stream_driver.hpp) against liblimo_hip.so, against the
    emulated C-ABI + the oracle's depth assignment (CPU tier), or with the ORACLE behind every C-ABI call (oracle=True)."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    csrc = os.path.join(_HERE, "..", "limo_amd", "csrc")
    out = os.path.join(_HERE, "cpp", "_build", "limo_stream_gpu" if gpu else "limo_stream_oracle" if oracle else "limo_stream_emu")
    if oracle:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", out, STREAM_APP_SRC] + SHIM_SRC + build_oracle_abi())
        return out
    if not gpu:
        oracle_dir = os.path.join(_HERE, "..", "oracle", "_build")
        abi = os.path.join(_HERE, "cpp", "_build", "libkba_emu_abi_depth.so")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-fPIC", "-shared", "-DKBA_EMU_EXPORT_ABI", "-DKBA_EMU_DEPTH", "-o", abi] + _SRC + [os.path.join(csrc, "host_misc.cpp"), "-L" + oracle_dir, "-loracle", "-Wl,-rpath," + os.path.abspath(oracle_dir)])
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", out, STREAM_APP_SRC] + SHIM_SRC + [abi, "-Wl,-rpath," + os.path.dirname(abi), "-L" + oracle_dir, "-loracle", "-Wl,-rpath," + os.path.abspath(oracle_dir)])
        return out
    libdir = os.path.join(_HERE, "..", "limo_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", out, STREAM_APP_SRC] + SHIM_SRC + ["-L" + libdir, "-llimo_hip", "-Wl,-rpath," + os.path.abspath(libdir)])
    return out


SHIM_TEST_ORACLE = os.path.join(_HERE, "cpp", "_build", "test_kba_shim_oracle")


def build_shim_tests(gpu=False, oracle=False):
    """tests/cpp/test_kba_shim.cpp + the kba shim, linked against the emulated C-ABI (CPU tier), liblimo_hip.so, or the oracle."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    csrc = os.path.join(_HERE, "..", "limo_amd", "csrc")
    test_src = os.path.join(_HERE, "cpp", "test_kba_shim.cpp")
    if oracle:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", SHIM_TEST_ORACLE, test_src] + SHIM_SRC + build_oracle_abi())
        return SHIM_TEST_ORACLE
    if not gpu:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-fPIC", "-shared", "-DKBA_EMU_EXPORT_ABI", "-o", ABI_LIB_PATH] + _SRC + [os.path.join(csrc, "host_misc.cpp")])
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", SHIM_TEST_EMU, test_src] + SHIM_SRC + [ABI_LIB_PATH, "-Wl,-rpath," + os.path.dirname(ABI_LIB_PATH)])
        return SHIM_TEST_EMU
    libdir = os.path.join(_HERE, "..", "limo_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", SHIM_TEST_GPU, test_src] + SHIM_SRC + ["-L" + libdir, "-llimo_hip", "-Wl,-rpath," + os.path.abspath(libdir)])
    return SHIM_TEST_GPU


def build_shim_tests_asan():
    """The shim tests, the shim and the emulated C-ABI in ONE executable under -fsanitize=address: the measurement tables of
    limo_amd/kba/keyframe.hpp hold pointers into a public std::map a caller may edit between two calls."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    out = os.path.join(_HERE, "cpp", "_build", "test_kba_shim_emu_asan")
    csrc = os.path.join(_HERE, "..", "limo_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-ffp-contract=off", "-std=c++17", "-pthread", "-DKBA_EMU_EXPORT_ABI",
                           "-o", out, os.path.join(_HERE, "cpp", "test_kba_shim.cpp")] + SHIM_SRC + _SRC + [os.path.join(csrc, "host_misc.cpp")])
    return out


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)


def solve_sharded(window, opts, n_shards, rank=0, world=1, allreduce=None, allgather=None):
    """Landmark-sharded emulation.  allreduce = None: n_shards virtual shards in this process; else this process is
    rank `rank` of `world` (shard s lives on rank s mod world) and
      allreduce(send_ndarray, recv_ndarray) must leave the element-wise sum over all ranks in recv (trimming, final landmarks),
      allgather(send_ndarray[count], recv_ndarray[world, count]) every rank's send, in rank order (the per-iteration exchange)."""
    lib = load()
    s = window.as_struct()
    rep = _ffi.BaReport()
    if allreduce is None:
        cb = C.cast(None, ALLREDUCE_FN)
    else:
        def _cb(send, recv, count, kind, _user):
            a = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_double)), shape=(count,))
            if kind == 2:
                b = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_double)), shape=(world, count))
                allgather(a, b)
            else:
                b = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_double)), shape=(count,))
                allreduce(a, b)  # send and recv may alias (in-place exchange)

        cb = ALLREDUCE_FN(_cb)
    rc = lib.emu_ba_solve_sharded(C.byref(s), C.byref(opts), int(n_shards), int(rank), int(world), cb, None, C.byref(rep))
    if rc != 0:
        raise RuntimeError("emu_ba_solve_sharded rc=%d" % rc)
    return rep.as_dict()


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB_PATH)
        dp, u8p = _ffi.c_double_p, _ffi.c_uint8_p
        lib.emu_ba_solve_batch.argtypes = [C.c_int32, C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), C.POINTER(_ffi.BaReport), C.c_int, C.POINTER(_ffi.SpeedPrior)]
        lib.emu_ba_evaluate.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), C.c_int, dp, dp, dp, dp, u8p]
        lib.emu_ba_solve_batch_streaming.argtypes = [C.c_int32, C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), C.POINTER(_ffi.BaReport), C.c_int]
        lib.emu_last_trimmed.argtypes = [C.c_int, _ffi.c_int32_p, C.c_int]
        lib.emu_ba_solve_sharded.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), C.c_int, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p, C.POINTER(_ffi.BaReport)]
        lib.emu_ba_evaluate_rows.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.SpeedPrior), C.c_int, C.POINTER(_ffi.BaOptions), C.c_int32, C.POINTER(_ffi.BaRow), _ffi.c_int32_p]
        _lib = lib
    return _lib


def evaluate_rows(window, opts, pose_only=False, prior=None):
    lib = load()
    s = window.as_struct()
    n = C.c_int32(0)
    cap = 64 + window.n_lm + 8 * window.n_kf
    rows = (_ffi.BaRow * cap)()
    rc = lib.emu_ba_evaluate_rows(C.byref(s), None if prior is None else C.byref(prior), int(pose_only), C.byref(opts), cap, rows, C.byref(n))
    if rc != 0:
        raise RuntimeError("emu_ba_evaluate_rows rc=%d" % rc)
    return _ffi.rows_as_dicts(rows, n.value)


def solve_batch(windows, opts, pose_only=False, prior=None):
    from limo_amd.window import struct_array

    lib = load()
    arr = struct_array(windows)
    reps = (_ffi.BaReport * len(windows))()
    rc = lib.emu_ba_solve_batch(len(windows), arr, C.byref(opts), reps, int(pose_only), None if prior is None else C.byref(prior))
    if rc != 0:
        raise RuntimeError("emu_ba_solve_batch rc=%d" % rc)
    return [r.as_dict() for r in reps]


def solve_batch_streaming(windows, opts, n_slots):
    """The batch through the streaming schedule (device-side scheduler emulated): n_slots windows in flight."""
    from limo_amd.window import struct_array

    lib = load()
    arr = struct_array(windows)
    reps = (_ffi.BaReport * len(windows))()
    rc = lib.emu_ba_solve_batch_streaming(len(windows), arr, C.byref(opts), reps, int(n_slots))
    if rc != 0:
        raise RuntimeError("emu_ba_solve_batch_streaming rc=%d" % rc)
    return [r.as_dict() for r in reps]


def last_trimmed(w=0):
    """Caller-order landmark indices trimmed in window w of the last solve_batch."""
    lib = load()
    n = lib.emu_last_trimmed(int(w), None, 0)
    out = np.zeros(max(1, n), np.int32)
    lib.emu_last_trimmed(int(w), out.ctypes.data_as(_ffi.c_int32_p), n)
    return np.sort(out[:n])


def evaluate(window, opts, apply_loss=True):
    lib = load()
    s = window.as_struct()
    M = window.n_obs
    cost = np.zeros(1)
    res = np.zeros((M, 3))
    jp = np.zeros((M, 3, 6))
    jl = np.zeros((M, 3, 3))
    valid = np.zeros(M, np.uint8)
    dp = lambda a: a.ctypes.data_as(_ffi.c_double_p)
    rc = lib.emu_ba_evaluate(C.byref(s), C.byref(opts), int(apply_loss), dp(cost), dp(res), dp(jp), dp(jl), valid.ctypes.data_as(_ffi.c_uint8_p))
    if rc != 0:
        raise RuntimeError("emu_ba_evaluate rc=%d" % rc)
    return float(cost[0]), res, jp, jl, valid


MATH_PROBE_LIB = os.path.join(_HERE, "cpp", "_build", "libmath_probe.so")


def build_math_probe(force=False):
    """tests/cpp/math_probe.hip: kba_math.hpp's reciprocal helpers as gfx950 kernels (tests/test_gpu_math.py holds them against
    IEEE arithmetic).  Cross-compiles without a GPU; the .so travels with the snapshot."""
    import shutil

    os.makedirs(os.path.dirname(MATH_PROBE_LIB), exist_ok=True)
    src = os.path.join(_HERE, "cpp", "math_probe.hip")
    deps = [src, os.path.join(_HERE, "..", "limo_amd", "csrc", "kba_math.hpp")]
    if not force and os.path.exists(MATH_PROBE_LIB) and all(os.path.getmtime(MATH_PROBE_LIB) >= os.path.getmtime(d) for d in deps):
        return MATH_PROBE_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", MATH_PROBE_LIB, src])
    return MATH_PROBE_LIB
