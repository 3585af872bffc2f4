"""Thin Python driver over the C-ABI (ctypes): context, batched solve, evaluate, pose-only.

Used by tests and bench.py.  It never computes anything itself: every call goes through
`limo_amd/lib/liblimo_hip.so` (include/limo_hip.h) and fails loudly if that library or a GPU is missing.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .window import struct_array


class LimoError(RuntimeError):
    pass


class NotEnoughKeyframes(LimoError):
    """BundleAdjusterKeyframes::NotEnoughKeyframesException (bundle_adjuster_keyframes.hpp:59-68)."""


def _check(rc, ctx=None, what=""):
    if rc == _ffi.LIMO_OK:
        return
    msg = ""
    if ctx is not None:
        msg = _ffi.load().limo_last_error(ctx).decode(errors="replace")
    if rc == _ffi.LIMO_ERR_NOT_ENOUGH_KF:
        raise NotEnoughKeyframes(msg or what)
    raise LimoError("%s failed: rc=%d %s" % (what, rc, msg))


class Context:
    def __init__(self, device=0, stream=None):
        self.lib = _ffi.load()
        self.device = device
        self.ptr = C.c_void_p()
        rc = self.lib.limo_ctx_create(device, C.byref(self.ptr))
        if rc != 0:
            raise LimoError("limo_ctx_create(device=%d) failed rc=%d (no gfx950 device? the product path has no CPU fallback)" % (device, rc))
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        _check(self.lib.limo_ctx_set_stream(self.ptr, C.c_void_p(stream)), self.ptr, "limo_ctx_set_stream")

    def close(self):
        if self.ptr:
            self.lib.limo_ctx_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- single window
    def solve(self, window, opts):
        s = window.as_struct()
        rep = _ffi.BaReport()
        _check(self.lib.limo_ba_solve(self.ptr, C.byref(s), C.byref(opts), C.byref(rep)), self.ptr, "limo_ba_solve")
        return rep.as_dict()

    def solve_sharded(self, window, opts, n_shards):
        """Landmark-sharded solve of one window: ranks of the communicator set by comm_init(), or
        n_shards virtual shards on this GPU when there is none."""
        s = window.as_struct()
        rep = _ffi.BaReport()
        _check(self.lib.limo_ba_solve_sharded(self.ptr, C.byref(s), C.byref(opts), int(n_shards), C.byref(rep)), self.ptr, "limo_ba_solve_sharded")
        return rep.as_dict()

    def exchange_stats(self):
        """limo_ctx_exchange_stats of the last solve_sharded: dict(exchanges, bytes, iterations)."""
        st = (C.c_int64 * 3)()
        _check(self.lib.limo_ctx_exchange_stats(self.ptr, st), self.ptr, "limo_ctx_exchange_stats")
        return {"exchanges": int(st[0]), "bytes": int(st[1]), "iterations": int(st[2])}

    def coop_fallbacks(self):
        """limo_ctx_coop_fallbacks: one-launch solves on this context that timed out at a barrier and were redone as launches."""
        return int(self.lib.limo_ctx_coop_fallbacks(self.ptr))

    def comm_init(self, unique_id, rank, world):
        _check(self.lib.limo_ctx_comm_init(self.ptr, unique_id, int(rank), int(world)), self.ptr, "limo_ctx_comm_init")

    def comm_init_host(self, rank, world, allgather, allreduce):
        """limo_ctx_comm_init_host: the exchange steps of solve_sharded through the caller's transport, staged in host memory.
        allgather(send[count], recv[world, count]) / allreduce(send[count], recv[count]) get numpy views of the staging buffers."""
        import numpy as np

        def _cb(send, recv, count, kind, _user):
            s = np.ctypeslib.as_array(send, shape=(count,))
            if kind == 0:
                allgather(s, np.ctypeslib.as_array(recv, shape=(world, count)))
            else:
                allreduce(s, np.ctypeslib.as_array(recv, shape=(count,)))

        self._xfn = _ffi.EXCHANGE_FN(_cb)  # (kept alive with the context)
        _check(self.lib.limo_ctx_comm_init_host(self.ptr, self._xfn, None, int(rank), int(world)), self.ptr, "limo_ctx_comm_init_host")

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        _check(self.lib.limo_comm_unique_id(buf), self.ptr, "limo_comm_unique_id")
        return buf.raw

    def landmark_init(self, ray_off, rays, use_depth):
        """limo_landmark_init: rays = ctypes array of _ffi.Ray (CSR by ray_off); returns (positions [n,3], ok [n])."""
        ray_off = np.ascontiguousarray(ray_off, np.int32)
        use_depth = np.ascontiguousarray(use_depth, np.uint8)
        n = len(use_depth)
        pos = np.zeros((n, 3))
        ok = np.zeros(n, np.uint8)
        rc = self.lib.limo_landmark_init(self.ptr, n, ray_off.ctypes.data_as(_ffi.c_int32_p), rays, use_depth.ctypes.data_as(_ffi.c_uint8_p),
                                         pos.ctypes.data_as(_ffi.c_double_p), ok.ctypes.data_as(_ffi.c_uint8_p))
        _check(rc, self.ptr, "limo_landmark_init")
        return pos, ok

    def adjust_pose_only(self, window, prior, opts):
        s = window.as_struct()
        rep = _ffi.BaReport()
        rc = self.lib.limo_ba_adjust_pose_only(self.ptr, C.byref(s), None if prior is None else C.byref(prior), C.byref(opts), C.byref(rep))
        _check(rc, self.ptr, "limo_ba_adjust_pose_only")
        return rep.as_dict()

    def evaluate(self, window, opts, apply_loss=True):
        s = window.as_struct()
        M = window.n_obs
        cost = np.zeros(1)
        res = np.zeros((M, 3))
        jp = np.zeros((M, 3, 6))
        jl = np.zeros((M, 3, 3))
        valid = np.zeros(M, np.uint8)
        dp = lambda a: a.ctypes.data_as(_ffi.c_double_p)
        rc = self.lib.limo_ba_evaluate(self.ptr, C.byref(s), C.byref(opts), int(apply_loss), dp(cost), dp(res), dp(jp), dp(jl), valid.ctypes.data_as(_ffi.c_uint8_p))
        _check(rc, self.ptr, "limo_ba_evaluate")
        return float(cost[0]), res, jp, jl, valid


    def evaluate_rows(self, window, opts, pose_only=False, prior=None):
        """limo_ba_evaluate_rows: the non-observation residual rows (ground plane, regularisers, speed prior) as dicts."""
        s = window.as_struct()
        n = C.c_int32(0)
        cap = 64 + window.n_lm + 8 * window.n_kf
        rows = (_ffi.BaRow * cap)()
        rc = self.lib.limo_ba_evaluate_rows(self.ptr, C.byref(s), None if prior is None else C.byref(prior), int(pose_only), C.byref(opts), cap, rows, C.byref(n))
        _check(rc, self.ptr, "limo_ba_evaluate_rows")
        assert n.value <= cap
        return _ffi.rows_as_dicts(rows, n.value)


class Batch:
    """Many independent windows resident in HBM (limo_ba_batch_*)."""

    def __init__(self, ctx, windows, arr=None):
        """arr: struct_array(windows) made earlier (the array of limo_ba_window a C / C++ caller holds anyway: building it from
        numpy windows is ~20 us of Python per window and not part of the library call)."""
        self.ctx = ctx
        self.lib = ctx.lib
        self.windows = list(windows)
        self._arr = struct_array(self.windows) if arr is None else arr
        self.ptr = C.c_void_p()
        _check(self.lib.limo_ba_batch_create(ctx.ptr, len(self.windows), self._arr, C.byref(self.ptr)), ctx.ptr, "limo_ba_batch_create")

    def solve(self, opts):
        _check(self.lib.limo_ba_batch_solve(self.ptr, C.byref(opts)), self.ctx.ptr, "limo_ba_batch_solve")

    def reset(self):
        _check(self.lib.limo_ba_batch_reset(self.ptr), self.ctx.ptr, "limo_ba_batch_reset")

    def download(self):
        """Writes optimised parameters into self.windows (in place) and returns the reports."""
        reps = (_ffi.BaReport * len(self.windows))()
        _check(self.lib.limo_ba_batch_download(self.ptr, self._arr, reps), self.ctx.ptr, "limo_ba_batch_download")
        return [r.as_dict() for r in reps]

    def trimmed(self, w=0):
        """Indices (caller's order) of the landmarks of window w removed by the trimming rounds of the last solve."""
        flags = np.zeros(max(1, self.windows[w].n_lm), np.uint8)
        _check(self.lib.limo_ba_batch_trimmed(self.ptr, int(w), flags.ctypes.data_as(_ffi.c_uint8_p)), self.ctx.ptr, "limo_ba_batch_trimmed")
        return np.nonzero(flags[: self.windows[w].n_lm])[0].astype(np.int32)

    def kernel_stats(self, reset=False):
        ms = C.c_double()
        n = C.c_int64()
        tot = C.c_double()
        sms, sn = C.c_double(), C.c_int64()
        _check(self.lib.limo_ba_batch_kernel_time(self.ptr, 1, C.byref(sms), C.byref(sn)), self.ctx.ptr, "limo_ba_batch_kernel_time")
        out = {"schur_ms": sms.value, "schur_launches": sn.value}
        _check(self.lib.limo_ba_batch_kernel_stats(self.ptr, int(reset), C.byref(ms), C.byref(n), C.byref(tot)), self.ctx.ptr, "limo_ba_batch_kernel_stats")
        out.update({"linearize_ms": ms.value, "linearize_launches": n.value, "total_ms": tot.value})
        return out

    def close(self):
        if self.ptr:
            self.lib.limo_ba_batch_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def evaluate_batch_time(ctx, windows, opts, reps=10):
    """limo_ba_evaluate_batch_time: device ms of one k_evaluate launch over `windows`; returns (ms, n_obs, n_depth_obs)."""
    arr = struct_array(windows)
    ms = C.c_double()
    _check(ctx.lib.limo_ba_evaluate_batch_time(ctx.ptr, len(windows), arr, C.byref(opts), int(reps), C.byref(ms)), ctx.ptr, "limo_ba_evaluate_batch_time")
    return ms.value, int(sum(w.n_obs for w in windows)), int(sum((w.obs_d > 0).sum() for w in windows))


def depth_default_params():
    p = _ffi.DepthParams()
    _ffi.load().limo_depth_default_params(C.byref(p))
    return p


def depth_estimate(ctx, frame, params=None, use_ground_labels=True):
    """limo_depth_estimate on one frame dict (see limo_amd.synth_lidar.make_frame): returns float32 depth per feature."""
    lib = ctx.lib
    p = params if params is not None else depth_default_params()
    cloud = np.ascontiguousarray(frame["cloud"], np.float32)
    uv = np.ascontiguousarray(frame["uv"], np.float32)
    T = np.ascontiguousarray(frame["T_cam_lidar"], np.float64)
    g = np.ascontiguousarray(frame["is_ground"], np.uint8) if use_ground_labels else None
    out = np.zeros(uv.shape[0], np.float32)
    rc = lib.limo_depth_estimate(
        ctx.ptr,
        cloud.ctypes.data_as(_ffi.c_float_p),
        cloud.shape[0],
        T.ctypes.data_as(_ffi.c_double_p),
        frame["f"],
        frame["cx"],
        frame["cy"],
        frame["w"],
        frame["h"],
        uv.ctypes.data_as(_ffi.c_float_p),
        uv.shape[0],
        None if g is None else g.ctypes.data_as(_ffi.c_uint8_p),
        C.byref(p),
        out.ctypes.data_as(_ffi.c_float_p),
    )
    _check(rc, ctx.ptr, "limo_depth_estimate")
    return out


def depth_estimate_begin(ctx, frame, params=None, use_ground_labels=True):
    """limo_depth_estimate_begin: the frame's depth assignment is enqueued on the context's stream; returns the handle
    depth_estimate_end wants (it keeps the cloud alive until then)."""
    p = params if params is not None else depth_default_params()
    cloud = np.ascontiguousarray(frame["cloud"], np.float32)
    uv = np.ascontiguousarray(frame["uv"], np.float32)
    T = np.ascontiguousarray(frame["T_cam_lidar"], np.float64)
    g = np.ascontiguousarray(frame["is_ground"], np.uint8) if use_ground_labels else None
    rc = ctx.lib.limo_depth_estimate_begin(
        ctx.ptr, cloud.ctypes.data_as(_ffi.c_float_p), cloud.shape[0], T.ctypes.data_as(_ffi.c_double_p), frame["f"], frame["cx"], frame["cy"],
        frame["w"], frame["h"], uv.ctypes.data_as(_ffi.c_float_p), uv.shape[0], None if g is None else g.ctypes.data_as(_ffi.c_uint8_p), C.byref(p))
    _check(rc, ctx.ptr, "limo_depth_estimate_begin")
    return (cloud, uv.shape[0])


def depth_estimate_end(ctx, handle):
    """limo_depth_estimate_end: float32 depth per feature of the frame given to depth_estimate_begin."""
    out = np.zeros(handle[1], np.float32)
    _check(ctx.lib.limo_depth_estimate_end(ctx.ptr, out.ctypes.data_as(_ffi.c_float_p), handle[1]), ctx.ptr, "limo_depth_estimate_end")
    return out


def depth_kernel_ms(ctx):
    """limo_depth_last_kernel_ms (after limo_depth_set_timing(ctx, 1)): dict of device milliseconds of the last launch group."""
    ms = np.zeros(4)
    _check(ctx.lib.limo_depth_last_kernel_ms(ctx.ptr, ms.ctypes.data_as(_ffi.c_double_p)), ctx.ptr, "limo_depth_last_kernel_ms")
    return {"k_project": ms[0], "ground_plane": ms[1], "k_features": ms[2], "total": ms[3]}


def depth_last_ground_plane(ctx, frame=0):
    """limo_depth_last_ground_plane: (RANSAC inliers, plane4) of a frame of the last depth call on this context."""
    pl = np.zeros(4)
    n = C.c_int32(0)
    rc = ctx.lib.limo_depth_last_ground_plane(ctx.ptr, frame, pl.ctypes.data_as(_ffi.c_double_p), C.byref(n))
    _check(rc, ctx.ptr, "limo_depth_last_ground_plane")
    return n.value, pl


def depth_estimate_batch(ctx, frames, params=None, use_ground_labels=True, device=False):
    """limo_depth_estimate_batch over a list of frame dicts of one rig (calibration of frames[0]).
    device=False: numpy clouds / features in, list of float32 depth arrays out.
    device=True: every frame dict carries torch CUDA tensors "cloud" (n,4 float32), "uv" (m,2 float32), optionally
    "is_ground" (m uint8); returns a list of torch CUDA float32 tensors (nothing crosses PCIe)."""
    lib = ctx.lib
    p = params if params is not None else depth_default_params()
    n = len(frames)
    arr = (_ffi.DepthFrame * max(1, n))()
    keep, outs = [], []
    for k, fr in enumerate(frames):
        if device:
            import torch

            cloud, uv = fr["cloud"].contiguous(), fr["uv"].contiguous()
            assert cloud.is_cuda and cloud.dtype == torch.float32 and uv.is_cuda and uv.dtype == torch.float32
            g = fr["is_ground"].contiguous() if use_ground_labels and fr.get("is_ground") is not None else None
            out = torch.empty(uv.shape[0], dtype=torch.float32, device=uv.device)
            arr[k] = _ffi.DepthFrame(cloud.data_ptr(), cloud.shape[0], uv.data_ptr(), uv.shape[0], g.data_ptr() if g is not None else None, out.data_ptr())
        else:
            cloud = np.ascontiguousarray(fr["cloud"], np.float32)
            uv = np.ascontiguousarray(fr["uv"], np.float32)
            g = np.ascontiguousarray(fr["is_ground"], np.uint8) if use_ground_labels else None
            out = np.zeros(uv.shape[0], np.float32)
            arr[k] = _ffi.DepthFrame(cloud.ctypes.data, cloud.shape[0], uv.ctypes.data, uv.shape[0], g.ctypes.data if g is not None else None, out.ctypes.data)
        keep.append((cloud, uv, g))
        outs.append(out)
    if n == 0:
        return []
    f0 = frames[0]
    T = np.ascontiguousarray(f0["T_cam_lidar"], np.float64)
    if device:
        import torch

        torch.cuda.current_stream().synchronize()  # the tensors above may still be in flight on torch's stream
    rc = lib.limo_depth_estimate_batch(ctx.ptr, n, arr, T.ctypes.data_as(_ffi.c_double_p), f0["f"], f0["cx"], f0["cy"], f0["w"], f0["h"], C.byref(p),
                                       _ffi.DEPTH_DEVICE_POINTERS if device else 0)
    _check(rc, ctx.ptr, "limo_depth_estimate_batch")
    return outs


def host_array(shape, dtype):
    """numpy array on page-locked host memory (limo_host_alloc); freed when the array is collected."""
    lib = _ffi.load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = lib.limo_host_alloc(max(1, n))
    if not ptr:
        raise MemoryError("limo_host_alloc(%d)" % n)

    class _Owner:
        def __del__(self, free=lib.limo_host_free, p=ptr):
            free(p)

    buf = (C.c_char * max(1, n)).from_address(ptr)
    buf._owner = _Owner()
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
