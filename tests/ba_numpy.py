"""Independent numpy statement of the keyframe-BA cost, for cross-checking oracle/ - no code
shared with oracle/ or limo_amd/csrc, vectorised instead of block by block, and usable with complex numbers (so that
scipy's complex-step Jacobian applies).  TEST INFRASTRUCTURE.

    P = Problem(window, options)              # wiring decided at the window's CURRENT parameters, like solve() does
    f = P.residuals(delta, removed=...)       # robustified residual vector: cost = 0.5 * f @ f
    P.cost(delta), P.apply(delta)             # delta: tangent displacement of the free parameters (see `layout`)

Reference lines behind every block are listed in SURVEY.md Appendix B; the ones restated here:
  reprojection / depth    internal/cost_functors_ceres.hpp:91-155,193-212, built bundle_adjuster_keyframes.cpp:578-620
  ground height           cost_functors_ceres.hpp:358-385, wiring bundle_adjuster_keyframes.cpp:517-562
  scale regularisation    cost_functors_ceres.hpp:229-241, bundle_adjuster_keyframes.cpp:704-716,890-904
  plane regularisers      cost_functors_ceres.hpp:399-428,512-518,533-547, bundle_adjuster_keyframes.cpp:769-818
  losses                  Ceres 1.13 loss_function.cc (ScaledLoss of CauchyLoss / HuberLoss / TrivialLoss)
  manifolds               bundle_adjuster_keyframes.cpp:181-193, internal/local_parameterizations.hpp:135-165
  constant blocks         bundle_adjuster_keyframes.cpp:198-219,722-728
"""
import numpy as np

FIX_POSE = 0  # enum limo_fixation { LIMO_FIX_POSE = 0, ... } (include/limo_hip.h), Keyframe::FixationStatus::Pose


def rot(q):
    """Eigen's toRotationMatrix polynomial WITHOUT normalising q; q[..., 4] -> [..., 3, 3]."""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3), dtype=q.dtype)
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - w * z)
    R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y)
    R[..., 2, 1] = 2 * (y * z + w * x)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def quat_mul(a, b):
    return np.stack([
        a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1] - a[..., 2] * b[..., 2] - a[..., 3] * b[..., 3],
        a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0] + a[..., 2] * b[..., 3] - a[..., 3] * b[..., 2],
        a[..., 0] * b[..., 2] - a[..., 1] * b[..., 3] + a[..., 2] * b[..., 0] + a[..., 3] * b[..., 1],
        a[..., 0] * b[..., 3] + a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1] + a[..., 3] * b[..., 0],
    ], axis=-1)


def pose_plus(pose, d):
    """ProductParameterization(QuaternionParameterization, Identity(3)): q <- exp(d[:3]) * q, t <- t + d[3:]."""
    th2 = (d[..., :3] * d[..., :3]).sum(-1)
    th = np.sqrt(th2)
    safe = np.where(th2 == 0, 1.0, th)
    s = np.where(th2 == 0, 1.0, np.sin(safe) / safe)
    dq = np.concatenate([np.where(th2 == 0, 1.0, np.cos(safe))[..., None], s[..., None] * d[..., :3]], axis=-1)
    return np.concatenate([quat_mul(dq, pose[..., :4]), pose[..., 4:] + d[..., 3:]], axis=-1)


def unit_plus(n, d):
    """FixScaleVectorPlus(1.0): (n + d) / |n + d|."""
    v = n + d
    return v / np.sqrt((v * v).sum(-1))[..., None]


def robustify(r, kind, a, weight):
    """rows r [m, k] of blocks with loss weight * {Trivial, Huber(a), Cauchy(a)}(s), s = |r|^2  ->  r * sqrt(rho(s) / s)."""
    s = (r * r).sum(-1)
    if kind == "trivial":
        rho = s
    elif kind == "huber":
        big = s.real > a * a
        rho = np.where(big, 2 * a * np.sqrt(np.where(big, s, 1.0)) - a * a, s)
    elif kind == "cauchy":
        rho = a * a * np.log(1 + s / (a * a))
    zero = s == 0
    fac = np.sqrt(weight * np.where(zero, 1.0, rho) / np.where(zero, 1.0, s))
    return r * fac[..., None]


class Problem:
    def __init__(self, w, o):
        self.w, self.o = w, o
        K = w.n_kf
        self.cam_R = rot(w.cam[:, 3:7])
        self.cam_t = w.cam[:, 7:10]
        # ---- ground-height rows: nearest keyframe with a plane (distance >= -10), weight 10 (1 - d / 25), none beyond 25 m
        gl = np.flatnonzero(w.lm_is_ground)
        usable = np.flatnonzero(w.kf_plane_dist >= -10.0)
        self.gp_lm, self.gp_kf, self.gp_w = np.zeros(0, int), np.zeros(0, int), np.zeros(0)
        if len(gl) and len(usable):
            R, t = rot(w.kf_pose[usable, :4]), w.kf_pose[usable, 4:]
            local = np.einsum("kij,lj->lki", R, w.lm_pos[gl]) + t[None]
            dist = np.sqrt((local * local).sum(-1))
            k = dist.argmin(1)
            dmin = dist[np.arange(len(gl)), k]
            keep = dmin < 25.0
            self.gp_lm, self.gp_kf, self.gp_w = gl[keep], usable[k[keep]], 10.0 * (1 - dmin[keep] / 25.0)
        self.n_depth = int((w.obs_d > 0).sum())
        self.n_gp = len(self.gp_lm)
        # ---- scale regularisation between the two oldest keyframes
        self.scale_w = None
        if K > 1:
            if self.n_depth > 10 or self.n_gp > 10:
                if self.n_gp < 30:
                    self.scale_w = 1000.0 / (self.n_depth + self.n_gp)
            else:
                self.scale_w = 1000.0
            self.scale0 = float(np.sqrt((self._rel_t(w.kf_pose[1], w.kf_pose[0]) ** 2).sum()))
        self.plane_regs = self.n_gp > 0 and K > 1
        # ---- which parameters move
        fixed = w.kf_fixation == FIX_POSE
        uses_plane = np.zeros(K, bool)
        uses_plane[self.gp_kf] = True
        if self.plane_regs:
            uses_plane[:] = True
        self.free_pose = ~fixed
        self.free_dir = ~fixed & uses_plane
        self.free_dist = ~fixed & uses_plane & (self.n_depth >= 10)
        seen = np.zeros(w.n_lm, bool)
        seen[w.obs_lm] = True
        self.lm_in_problem = seen
        self._layout()

    @staticmethod
    def _rel_t(pa, pb):
        """translation of T(pa) * T(pb)^-1 = t_a - R_a R_b^T t_b"""
        Ra, Rb = rot(pa[..., :4]), rot(pb[..., :4])
        return pa[..., 4:] - np.einsum("...ij,...kj,...k->...i", Ra, Rb, pb[..., 4:])

    def _layout(self, removed=()):
        w = self.w
        self.removed = np.zeros(w.n_lm, bool)
        self.removed[list(removed)] = True
        self.free_lm = self.lm_in_problem & ~self.removed
        self.i_pose = np.flatnonzero(self.free_pose)
        self.i_dir = np.flatnonzero(self.free_dir)
        self.i_dist = np.flatnonzero(self.free_dist)
        self.i_lm = np.flatnonzero(self.free_lm)
        self.n_free = 6 * len(self.i_pose) + 3 * len(self.i_dir) + len(self.i_dist) + 3 * len(self.i_lm)

    def remove(self, landmarks):
        """trimming: every block of these landmarks leaves the problem (robust_solving.cpp:196-214)"""
        self._layout(landmarks)

    def params(self, delta=None):
        """(pose, plane_dir, plane_dist, lm) at Plus(current, delta)"""
        w = self.w
        pose, ndir, dist, lm = w.kf_pose, w.kf_plane_dir, w.kf_plane_dist, w.lm_pos
        if delta is None:
            return pose, ndir, dist, lm
        delta = np.asarray(delta)
        dt = np.result_type(delta.dtype, np.float64)
        pose, ndir, dist, lm = pose.astype(dt), ndir.astype(dt), dist.astype(dt), lm.astype(dt)
        o = 0
        n = 6 * len(self.i_pose)
        pose[self.i_pose] = pose_plus(pose[self.i_pose], delta[o:o + n].reshape(-1, 6))
        o += n
        n = 3 * len(self.i_dir)
        ndir[self.i_dir] = unit_plus(ndir[self.i_dir], delta[o:o + n].reshape(-1, 3))
        o += n
        n = len(self.i_dist)
        dist[self.i_dist] = dist[self.i_dist] + delta[o:o + n]
        o += n
        lm[self.i_lm] = lm[self.i_lm] + delta[o:].reshape(-1, 3)
        return pose, ndir, dist, lm

    def apply(self, delta):
        pose, ndir, dist, lm = self.params(delta)
        w = self.w
        w.kf_pose[:], w.kf_plane_dir[:], w.kf_plane_dist[:], w.lm_pos[:] = pose.real, ndir.real, dist.real, lm.real

    def residuals(self, delta=None):
        w, o = self.w, self.o
        pose, ndir, dist, lm = self.params(delta)
        R, t = rot(pose[:, :4]), pose[:, 4:]
        out = []
        # ---- reprojection + depth
        keep = ~self.removed[w.obs_lm]
        k, l, c = w.obs_kf[keep], w.obs_lm[keep], w.obs_cam[keep]
        x = np.einsum("nij,nj->ni", R[k], lm[l]) + t[k]
        y = np.einsum("nij,nj->ni", self.cam_R[c], x) + self.cam_t[c]
        if (np.abs(y[:, 2].real) < 0.01).any():
            raise FloatingPointError("a landmark within 1 cm of a camera plane: the reference's functor returns false")
        f, cx, cy = w.cam[c, 0], w.cam[c, 1], w.cam[c, 2]
        r = np.stack([f * y[:, 0] / y[:, 2] + cx - w.obs_u[keep].astype(np.float64), f * y[:, 1] / y[:, 2] + cy - w.obs_v[keep].astype(np.float64)], axis=1)
        out.append(robustify(r, "cauchy", o.reprojection_thres, w.lm_weight[l]).ravel())
        has_d = w.obs_d[keep] > 0
        rd = (y[has_d, 2] - w.obs_d[keep][has_d].astype(np.float64))[:, None]
        out.append(robustify(rd, "cauchy", o.depth_thres, w.lm_weight[l[has_d]]).ravel())
        # ---- ground height
        g = ~self.removed[self.gp_lm]
        gl, gk = self.gp_lm[g], self.gp_kf[g]
        xg = np.einsum("nij,nj->ni", R[gk], lm[gl]) + t[gk]
        rg = ((ndir[gk] * xg).sum(1) + dist[gk])[:, None]
        out.append(robustify(rg, "huber", 0.1, self.gp_w[g]).ravel())
        # ---- regularisers
        if self.scale_w is not None:
            d = self._rel_t(pose[1], pose[0])
            out.append(robustify(np.array([[np.sqrt((d * d).sum()) - self.scale0]]), "trivial", 0, self.scale_w).ravel())
        if self.plane_regs:
            out.append(robustify(ndir[1:] - ndir[:-1], "trivial", 0, 30.0).ravel())
            out.append(robustify((dist[1:] - dist[:-1])[:, None], "trivial", 0, 10.0).ravel())
            d = self._rel_t(pose[:-1], pose[1:])
            z = (d * d).sum(1)
            d = np.where((z.real > 0)[:, None], d / np.sqrt(np.where(z.real > 0, z, 1.0))[:, None], d)
            out.append(robustify((ndir[:-1] * d).sum(1)[:, None], "trivial", 0, 20.0).ravel())
            out.append(robustify(np.array([0.0, 0.0, 1.0]) - ndir, "trivial", 0, 10.0).ravel())
        return np.concatenate(out)

    def sparsity(self):
        """scipy.sparse pattern (rows of `residuals` x free parameters): which tangent coordinates a row can depend on."""
        import scipy.sparse as sp

        w = self.w
        col_pose = np.full(w.n_kf, -1)
        col_pose[self.i_pose] = 6 * np.arange(len(self.i_pose))
        o = 6 * len(self.i_pose)
        col_dir = np.full(w.n_kf, -1)
        col_dir[self.i_dir] = o + 3 * np.arange(len(self.i_dir))
        o += 3 * len(self.i_dir)
        col_dist = np.full(w.n_kf, -1)
        col_dist[self.i_dist] = o + np.arange(len(self.i_dist))
        o += len(self.i_dist)
        col_lm = np.full(w.n_lm, -1)
        col_lm[self.i_lm] = o + 3 * np.arange(len(self.i_lm))
        rows, cols = [], []
        n_rows = 0

        def add(nrow, first_cols_widths):
            """block of nrow[i] = const rows per item; first_cols_widths: list of (first column per item or -1, width)"""
            nonlocal n_rows
            n_items, per = nrow
            base = n_rows + per * np.arange(n_items)
            for first, width in first_cols_widths:
                ok = first >= 0
                for r in range(per):
                    for c in range(width):
                        rows.append(base[ok] + r)
                        cols.append(first[ok] + c)
            n_rows += per * n_items

        keep = ~self.removed[w.obs_lm]
        k, l = w.obs_kf[keep], w.obs_lm[keep]
        add((len(k), 2), [(col_pose[k], 6), (col_lm[l], 3)])
        has_d = w.obs_d[keep] > 0
        add((int(has_d.sum()), 1), [(col_pose[k[has_d]], 6), (col_lm[l[has_d]], 3)])
        g = ~self.removed[self.gp_lm]
        gl, gk = self.gp_lm[g], self.gp_kf[g]
        add((len(gl), 1), [(col_pose[gk], 6), (col_dir[gk], 3), (col_dist[gk], 1), (col_lm[gl], 3)])
        one = lambda v: np.array([v])
        if self.scale_w is not None:
            add((1, 1), [(one(col_pose[1]), 6), (one(col_pose[0]), 6)])
        if self.plane_regs:
            a, b = np.arange(w.n_kf - 1), np.arange(1, w.n_kf)
            add((len(a), 3), [(col_dir[a], 3), (col_dir[b], 3)])
            add((len(a), 1), [(col_dist[a], 1), (col_dist[b], 1)])
            add((len(a), 1), [(col_pose[a], 6), (col_pose[b], 6), (col_dir[a], 3)])
            add((w.n_kf, 3), [(col_dir[np.arange(w.n_kf)], 3)])
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        return sp.csr_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n_rows, self.n_free))

    def cost(self, delta=None):
        f = self.residuals(delta)
        return 0.5 * float((f * f).sum().real)
