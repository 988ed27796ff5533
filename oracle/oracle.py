"""CPU oracle for the DCS pose-graph path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (libpgo.so + toy-robust-backend-slam_amd) never does.

PARITY UNPINNED: the reference ships no tests / golden vectors for this path and
cannot be built here (Ceres, Eigen3, Boost absent).  See oracle/pgo_oracle.c.

Contents (paths relative to /root/reference/DCS-ceres):
  read_g2o()        ReadG2O::ReadG2O            include/g2o_util.h:23-89
  glibc_rand()      the C library rand() the reference's injector calls
  add_random_C()    ReadG2O::add_random_C       include/g2o_util.h:151-171
  evaluate()        residual blocks + Huber     src/ceres_error.cpp:42-94,135-196, main.cpp:66-68
  lm_direct()       ceres::Solve, default options, SPARSE_NORMAL_CHOLESKY (main.cpp:154-163):
                    Ceres 2.x TrustRegionMinimizer + LevenbergMarquardtStrategy policy with the
                    linear system solved by a sparse direct factorisation (scipy SuperLU)
  lm_pcg()          the same policy with block-Jacobi PCG, in C (pgo_oracle_lm_pcg): the "port"
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_VARIANTS = {}

# compiler flags of the CPU-baseline variants bench.py times next to the default build (oracle/Makefile: -O2 -fopenmp)
VARIANT_FLAGS = {"O3-native": "-O3 -march=native -fopenmp -fPIC -std=c11 -D_POSIX_C_SOURCE=200809L",
                 "O0": "-O0 -fopenmp -fPIC -std=c11 -D_POSIX_C_SOURCE=200809L"}


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libpgo_oracle.so")
    src = os.path.join(_HERE, "pgo_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpgo_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_variant(name: str) -> str:
    """the same source with other compiler flags (VARIANT_FLAGS), built where it runs (-march=native): libpgo_oracle_<name>.so"""
    tag = name
    if "native" in VARIANT_FLAGS[name]:   # machine-specific code: never reuse a build from another CPU (the tree travels to the GPU box)
        import hashlib
        try:
            model = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
        except OSError:
            model = []
        tag += "_" + hashlib.sha256("".join(model).encode()).hexdigest()[:10]
    so = os.path.join(_HERE, "libpgo_oracle_%s.so" % tag)
    src = os.path.join(_HERE, "pgo_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc"] + VARIANT_FLAGS[name].split() + ["-shared", "-o", so, src, "-lm"])
    return so


class _Opts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("method", "max_iters", "fixed_pose", "jacobi_scaling")] + \
               [(n, C.c_double) for n in ("phi", "huber_delta", "ftol", "gtol", "ptol", "radius0", "max_radius",
                                          "min_radius", "min_relative_decrease", "min_lm_diagonal",
                                          "max_lm_diagonal", "pcg_rtol")] + \
               [(n, C.c_int32) for n in ("pcg_max_iters", "threads", "verbose", "block_poses", "chain_len", "_pad")]


class _Iter(C.Structure):
    _fields_ = [("iter", C.c_int32), ("step_ok", C.c_int32)] + \
               [(n, C.c_double) for n in ("cost", "cost_change", "gradient_max_norm", "step_norm",
                                          "relative_decrease", "radius")] + \
               [("pcg_iters", C.c_int32), ("_pad", C.c_int32), ("pcg_rel_residual", C.c_double),
                ("seconds", C.c_double)]


class _Summary(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("termination", "iterations", "successful_steps", "total_pcg_iters")] + \
               [(n, C.c_double) for n in ("initial_cost", "final_cost", "seconds_total", "seconds_eval",
                                          "seconds_assemble", "seconds_linear", "seconds_candidate")]


def lib(variant=None):
    """the oracle library; variant = a key of VARIANT_FLAGS selects the same source built with those flags"""
    global _LIB
    if variant is not None:
        if variant not in _VARIANTS:
            _VARIANTS[variant] = _bind(C.CDLL(build_variant(variant)))
        return _VARIANTS[variant]
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


def _bind(L):
    dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.pgo_oracle_edge.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp, dp]
    L.pgo_oracle_edge.restype = None
    L.pgo_oracle_huber.argtypes = [C.c_double, C.c_double, dp]
    L.pgo_oracle_huber.restype = None
    L.pgo_oracle_edge_w.argtypes = [dp, dp, dp, dp, C.c_int, C.c_double, dp, dp]
    L.pgo_oracle_edge_w.restype = None
    L.pgo_oracle_edge_chi2.argtypes = [C.c_int, dp, C.c_int, ip, ip, dp, dp, dp]
    L.pgo_oracle_edge_chi2.restype = None
    L.pgo_oracle_eval_w.argtypes = [C.c_int, dp, C.c_int, ip, ip, dp, dp, bp, C.c_int, C.c_double, C.c_double,
                                    C.c_int, dp, dp, C.c_int]
    L.pgo_oracle_eval_w.restype = C.c_double
    L.pgo_oracle_eval_sc.argtypes = [C.c_int, dp, C.c_int, ip, ip, dp, bp, dp, C.c_double, C.c_double, C.c_int, dp, dp,
                                     dp, dp, C.c_int]
    L.pgo_oracle_eval_sc.restype = C.c_double
    L.pgo_oracle_lm_pcg_w.argtypes = [C.c_int, dp, C.c_int, ip, ip, dp, dp, bp, C.POINTER(_Opts), C.POINTER(_Iter),
                                      C.c_int, C.POINTER(C.c_int), C.POINTER(_Summary)]
    L.pgo_oracle_lm_pcg_w.restype = C.c_int
    L.pgo_oracle_normal_eq_w.argtypes = [C.c_int, dp, C.c_int, ip, ip, dp, dp, bp, C.c_int, C.c_double, C.c_double,
                                         C.c_int, dp, dp, dp, dp, dp, C.c_int]
    L.pgo_oracle_normal_eq_w.restype = C.c_int
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _bp(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


# --------------------------------------------------------------------- graph
@dataclass
class Graph:
    pose_id: np.ndarray          # int32 [N]
    poses: np.ndarray            # float64 [N,3]
    ia: np.ndarray               # int32 [E]  (positions, as the reference indexes nNodes[])
    ib: np.ndarray
    meas: np.ndarray             # float64 [E,3]
    info: np.ndarray             # float64 [E,6]
    kind: np.ndarray             # uint8 [E]  0 odometry, 1 closure, 2 bogus

    @property
    def n_poses(self):
        return len(self.pose_id)

    @property
    def n_edges(self):
        return len(self.ia)

    def copy(self):
        return Graph(*(np.array(getattr(self, f), copy=True) for f in
                       ("pose_id", "poses", "ia", "ib", "meas", "info", "kind")))


def read_g2o(path: str) -> Graph:
    """include/g2o_util.h:23-89: split on single spaces (compressed), tags VERTEX_SE2|VERTEX2 and
    EDGE_SE2|EDGE2, odometry iff abs(a-b) < 5, edge lists in file order: odometry then closure."""
    ids, poses = [], []
    groups = ([], [])
    with open(path, "r") as f:
        for line in f:
            w = [t for t in line.rstrip("\n").split(" ")]
            # boost::split with token_compress_on keeps one empty token at a leading space
            if w and w[0] == "" and len(w) > 1:
                continue
            w = [t for t in w if t != ""]
            if not w:
                continue
            if w[0] in ("VERTEX_SE2", "VERTEX2"):
                ids.append(int(w[1]))
                poses.append((float(w[2]), float(w[3]), float(w[4])))
            elif w[0] in ("EDGE_SE2", "EDGE2"):
                a, b = int(w[1]), int(w[2])
                rec = (a, b) + tuple(float(t) for t in w[3:12])
                groups[0 if abs(a - b) < 5 else 1].append(rec)
    rows = groups[0] + groups[1]
    arr = np.array(rows, dtype=np.float64).reshape(-1, 11)
    kind = np.array([0] * len(groups[0]) + [1] * len(groups[1]), dtype=np.uint8)
    return Graph(np.array(ids, np.int32), np.array(poses, np.float64).reshape(-1, 3),
                 arr[:, 0].astype(np.int32), arr[:, 1].astype(np.int32),
                 np.ascontiguousarray(arr[:, 2:5]), np.ascontiguousarray(arr[:, 5:11]), kind)


class GlibcRand:
    """glibc's rand()/srand() (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]),
    restated so that the injector's draws can be checked without calling the C library."""
    RAND_MAX = 2147483647

    def __init__(self, seed: int):
        seed = seed & 0xFFFFFFFF
        if seed == 0:
            seed = 1
        r = [0] * 34
        r[0] = seed
        for i in range(1, 31):
            # 16807 * r[i-1] % 2147483647 computed with Schrage's trick on signed 32-bit words
            word = r[i - 1] if r[i - 1] < 2 ** 31 else r[i - 1] - 2 ** 32
            hi = word // 127773 if word >= 0 else -((-word) // 127773)
            lo = word - hi * 127773
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r[i] = word
        for i in range(31, 34):
            r[i] = r[i - 31]
        self.r = r
        for _ in range(34, 344):
            self._step()

    def _step(self):
        r = self.r
        v = (r[-31] + r[-3]) & 0xFFFFFFFF
        r.append(v)
        if len(r) > 64:
            del r[:len(r) - 34]
        return v

    def rand(self) -> int:
        return self._step() >> 1


def add_random_C(g: Graph, count: int, seed: int) -> Graph:
    """include/g2o_util.h:151-171 with srand(seed): a, b, then three rand()/RAND_MAX integer divisions."""
    rng = GlibcRand(seed)
    n = g.n_poses
    ia, ib, meas = [], [], []
    for _ in range(count):
        a = rng.rand() % n
        b = rng.rand() % n
        if a == b:
            b = (b + 1) % n
        m = [float(rng.rand() // GlibcRand.RAND_MAX) for _ in range(3)]
        ia.append(a)
        ib.append(b)
        meas.append(m)
    out = g.copy()
    if count:
        out.ia = np.concatenate([g.ia, np.array(ia, np.int32)])
        out.ib = np.concatenate([g.ib, np.array(ib, np.int32)])
        out.meas = np.concatenate([g.meas, np.array(meas, np.float64).reshape(-1, 3)])
        out.info = np.concatenate([g.info, np.tile(np.array([2.0, 0, 0, 300.0, 0, 300.0]), (count, 1))])
        out.kind = np.concatenate([g.kind, np.full(count, 2, np.uint8)])
    return out


# ---------------------------------------------------------------- evaluation
def edge(P1, P2, meas, dcs: bool, phi: float = 0.5, jac: bool = True, info=None):
    """one residual block; info = (I11, I12, I13, I22, I23, I33) switches to the whitened / chi2-DCS form"""
    P1 = np.ascontiguousarray(P1, np.float64)
    P2 = np.ascontiguousarray(P2, np.float64)
    m = np.ascontiguousarray(meas, np.float64)
    e = np.zeros(3)
    J = np.zeros(18) if jac else None
    if info is None:
        lib().pgo_oracle_edge(_dp(P1), _dp(P2), _dp(m), int(dcs), phi, _dp(e), _dp(J))
    else:
        w = np.ascontiguousarray(info, np.float64)
        lib().pgo_oracle_edge_w(_dp(P1), _dp(P2), _dp(m), _dp(w), int(dcs), phi, _dp(e), _dp(J))
    return (e, J.reshape(3, 6)) if jac else e


def _info(g, info_weighting):
    return np.ascontiguousarray(g.info, np.float64) if info_weighting else None


def edge_chi2(g: "Graph", poses=None):
    """compute_edge_mahalanobis (src/layer_manager.cpp:230-282) for every edge: r' Omega r of the plain residual"""
    poses = np.ascontiguousarray(g.poses if poses is None else poses, np.float64)
    ia, ib = np.ascontiguousarray(g.ia, np.int32), np.ascontiguousarray(g.ib, np.int32)
    meas, w = np.ascontiguousarray(g.meas, np.float64), np.ascontiguousarray(g.info, np.float64)
    out = np.zeros(g.n_edges)
    lib().pgo_oracle_edge_chi2(g.n_poses, _dp(poses), g.n_edges, _ip(ia), _ip(ib), _dp(meas), _dp(w), _dp(out))
    return out


def huber(s: float, delta: float = 0.01):
    rho = np.zeros(3)
    lib().pgo_oracle_huber(s, delta, _dp(rho))
    return rho


def evaluate(g: Graph, poses=None, method: int = 1, phi: float = 0.5, delta: float = 0.01, apply_loss: bool = True,
             want_r: bool = True, want_J: bool = True, threads: int = 1, info_weighting: bool = False):
    poses = np.ascontiguousarray(g.poses if poses is None else poses, np.float64)
    E = g.n_edges
    w = _info(g, info_weighting)
    r = np.zeros((E, 3)) if want_r else None
    J = np.zeros((E, 18)) if want_J else None
    ia, ib = np.ascontiguousarray(g.ia, np.int32), np.ascontiguousarray(g.ib, np.int32)
    meas, kind = np.ascontiguousarray(g.meas, np.float64), np.ascontiguousarray(g.kind, np.uint8)
    cost = lib().pgo_oracle_eval_w(g.n_poses, _dp(poses), E, _ip(ia), _ip(ib), _dp(meas), _dp(w), _bp(kind), method, phi,
                                   delta, int(apply_loss), _dp(r), _dp(J), threads)
    return cost, r, J


def evaluate_sc(g: Graph, poses=None, switches=None, lam: float = 1.0, delta: float = 0.01, apply_loss: bool = True,
                want: bool = True, threads: int = 1):
    """METHOD 2 residual blocks.  switches: [E] (entries of odometry edges ignored; default all 1).
    Returns cost, r [E,3], J [E,18], Js [E,3], q [E] (the four arrays are None when want is False)."""
    poses = np.ascontiguousarray(g.poses if poses is None else poses, np.float64)
    E = g.n_edges
    sw = np.ascontiguousarray(np.ones(E) if switches is None else switches, np.float64)
    r, J, Js, q = (np.zeros((E, 3)), np.zeros((E, 18)), np.zeros((E, 3)), np.zeros(E)) if want else (None, None, None, None)
    ia, ib = np.ascontiguousarray(g.ia, np.int32), np.ascontiguousarray(g.ib, np.int32)
    meas, kind = np.ascontiguousarray(g.meas, np.float64), np.ascontiguousarray(g.kind, np.uint8)
    cost = lib().pgo_oracle_eval_sc(g.n_poses, _dp(poses), E, _ip(ia), _ip(ib), _dp(meas), _bp(kind), _dp(sw), lam, delta,
                                    int(apply_loss), _dp(r), _dp(J), _dp(Js), _dp(q), threads)
    return cost, r, J, Js, q


def normal_eq(g: Graph, poses=None, method=1, phi=0.5, delta=0.01, fixed_pose=0, s=None, x=None, threads=1,
              info_weighting=False):
    """(gradient S J'r, diagonal blocks of (JS)'(JS), optional y = (JS)'(JS) x) at `poses`."""
    poses = np.ascontiguousarray(g.poses if poses is None else poses, np.float64)
    N = g.n_poses
    gout, hd = np.zeros(3 * N), np.zeros((N, 9))
    y = np.zeros(3 * N) if x is not None else None
    xs = np.ascontiguousarray(x, np.float64) if x is not None else None
    ss = np.ascontiguousarray(s, np.float64) if s is not None else None
    ia, ib = np.ascontiguousarray(g.ia, np.int32), np.ascontiguousarray(g.ib, np.int32)
    meas, kind = np.ascontiguousarray(g.meas, np.float64), np.ascontiguousarray(g.kind, np.uint8)
    w = _info(g, info_weighting)
    lib().pgo_oracle_normal_eq_w(N, _dp(poses), g.n_edges, _ip(ia), _ip(ib), _dp(meas), _dp(w), _bp(kind), method, phi,
                                 delta, fixed_pose, _dp(ss), _dp(gout), _dp(hd), _dp(xs), _dp(y), threads)
    return gout, hd, y


# ------------------------------------------------------------------ LM policy
@dataclass
class Options:
    method: int = 1
    max_iters: int = 50
    fixed_pose: int = 0
    jacobi_scaling: int = 1
    phi: float = 0.5
    huber_delta: float = 0.01
    ftol: float = 1e-6
    gtol: float = 1e-10
    ptol: float = 1e-8
    radius0: float = 1e4
    max_radius: float = 1e16
    min_radius: float = 1e-32
    min_relative_decrease: float = 1e-3
    min_lm_diagonal: float = 1e-6
    max_lm_diagonal: float = 1e32
    pcg_rtol: float = 1e-12
    pcg_max_iters: int = 100000
    threads: int = 1
    verbose: int = 0
    pcg_block_poses: int = 1   # lm_pcg only: poses per block-Jacobi block
    pcg_chain_len: int = 0     # lm_pcg only: > 0 = chain (block-tridiagonal) preconditioner over segments of this many poses
    info_weighting: int = 0    # 1: whitened residuals + chi2 DCS (optional mode, see pgo_oracle.c edge_functor_jet)


@dataclass
class Result:
    poses: np.ndarray
    termination: int
    iterations: int
    successful_steps: int
    initial_cost: float
    final_cost: float
    records: list = field(default_factory=list)
    seconds: dict = field(default_factory=dict)
    total_pcg_iters: int = 0


TERM = {1: "CONVERGENCE_FTOL", 2: "CONVERGENCE_GTOL", 3: "CONVERGENCE_PTOL", 4: "NO_CONVERGENCE", 5: "MIN_RADIUS",
        6: "FAILURE"}


def lm_direct(g: Graph, opt: Options = Options()) -> Result:
    """Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy, defaults of main.cpp:154-163, with the
    normal equations (J'J + D'D) y = J'r solved by a sparse direct factorisation."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    N, E = g.n_poses, g.n_edges
    x = np.array(g.poses, np.float64, copy=True)
    fixed = opt.fixed_pose
    free = np.ones(3 * N, bool)
    if fixed >= 0:
        free[3 * fixed:3 * fixed + 3] = False
    free_idx = np.nonzero(free)[0]
    col_of = -np.ones(3 * N, np.int64)
    col_of[free_idx] = np.arange(len(free_idx))
    # sparse pattern of J (3E x 3N): rows 3e+i, cols 3*ia+k / 3*ib+k
    rows = np.repeat(np.arange(3 * E).reshape(E, 3), 6, axis=1).reshape(-1)
    cols = np.concatenate([3 * g.ia[:, None] + np.arange(3), 3 * g.ib[:, None] + np.arange(3)], axis=1)
    cols = np.tile(cols, (1, 3)).reshape(-1).astype(np.int64)

    def jac_matrix(J):
        A = sp.csr_matrix((J.reshape(-1), (rows, cols)), shape=(3 * E, 3 * N))
        return A[:, free_idx].tocsc()

    def ev(p, with_j):
        return evaluate(g, p, opt.method, opt.phi, opt.huber_delta, True, with_j, with_j, opt.threads,
                        bool(opt.info_weighting))

    t_start = time.perf_counter()
    tm = dict(eval=0.0, linear=0.0, candidate=0.0)
    t0 = time.perf_counter()
    cost, r, J = ev(x, True)
    tm["eval"] += time.perf_counter() - t0
    res = Result(x, 4, 0, 0, cost, cost)
    if not np.isfinite(cost):
        res.termination = 6
        return res
    A = jac_matrix(J)
    rvec = r.reshape(-1)
    if opt.jacobi_scaling:
        s = 1.0 / (1.0 + np.sqrt(np.asarray(A.multiply(A).sum(axis=0)).reshape(-1)))
    else:
        s = np.ones(A.shape[1])
    grad = A.T @ rvec
    gmax = float(np.max(np.abs(grad))) if len(grad) else 0.0
    x_norm = float(np.linalg.norm(x.reshape(-1)[free_idx]))
    radius, dec = opt.radius0, 2.0
    prev_success, invalid_run = True, 0
    recs = [dict(iter=0, step_ok=1, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0,
                 relative_decrease=0.0, radius=radius)]
    it = 0
    term = 4
    while True:
        it += 1
        if it > opt.max_iters:
            term = 4
            it -= 1
            break
        if prev_success and gmax <= opt.gtol:
            term = 2
            it -= 1
            break
        if radius < opt.min_radius:
            term = 5
            it -= 1
            break
        As = A @ sp.diags(s)
        H = (As.T @ As).tocsc()
        diag = np.clip(H.diagonal(), opt.min_lm_diagonal, opt.max_lm_diagonal)
        D2 = diag / radius
        gs = s * grad
        t0 = time.perf_counter()
        y = spla.splu((H + sp.diags(D2)).tocsc()).solve(gs)
        tm["linear"] += time.perf_counter() - t0
        m = As @ (-y)
        model = float(-m @ (rvec + 0.5 * m))
        rec = dict(iter=it, step_ok=0, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0,
                   relative_decrease=0.0, radius=radius)
        if not np.all(np.isfinite(y)) or not (model > 0.0):
            invalid_run += 1
            if invalid_run >= 5:
                term = 6
                break
            radius /= dec
            dec *= 2.0
            prev_success = False
            rec.update(step_ok=-1, radius=radius)
            recs.append(rec)
            continue
        invalid_run = 0
        delta = np.zeros(3 * N)
        delta[free_idx] = -s * y
        cand = x + delta.reshape(N, 3)
        t0 = time.perf_counter()
        cand_cost, _, _ = ev(cand, False)
        tm["candidate"] += time.perf_counter() - t0
        if not np.isfinite(cand_cost):
            cand_cost = np.finfo(np.float64).max
        step_norm = float(np.linalg.norm(delta))
        cost_change = cost - cand_cost
        rec.update(step_norm=step_norm, cost_change=cost_change)
        if step_norm <= opt.ptol * (x_norm + opt.ptol):
            term = 3
            recs.append(rec)
            break
        if abs(cost_change) <= opt.ftol * cost:
            term = 1
            recs.append(rec)
            break
        rho = cost_change / model if cand_cost < np.finfo(np.float64).max else -np.inf
        rec.update(relative_decrease=rho)
        if rho > opt.min_relative_decrease:
            x = cand
            x_norm = float(np.linalg.norm(x.reshape(-1)[free_idx]))
            t0 = time.perf_counter()
            cost, r, J = ev(x, True)
            tm["eval"] += time.perf_counter() - t0
            if not np.isfinite(cost):
                term = 6
                break
            A = jac_matrix(J)
            rvec = r.reshape(-1)
            grad = A.T @ rvec
            gmax = float(np.max(np.abs(grad)))
            radius = min(opt.max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            dec = 2.0
            prev_success = True
            res.successful_steps += 1
            rec.update(step_ok=1, cost=cost, gradient_max_norm=gmax)
        else:
            radius /= dec
            dec *= 2.0
            prev_success = False
            rec.update(step_ok=0, cost=cand_cost)
        rec.update(radius=radius)
        recs.append(rec)
        if opt.verbose:
            print("%4d % .6e % .2e % .2e % .2e % .2e % .2e" % (it, rec["cost"], cost_change, gmax, step_norm, rho,
                                                               radius))
    res.poses = x
    res.termination = term
    res.iterations = it
    res.final_cost = cost
    res.records = recs
    tm["total"] = time.perf_counter() - t_start
    res.seconds = tm
    return res


def lm_pcg(g: Graph, opt: Options = Options(), variant=None) -> Result:
    """Same LM policy, linear solve by block-Jacobi PCG, all in C (OpenMP): the CPU "port".
    variant: a key of VARIANT_FLAGS = the same source built with other compiler flags (bench.py's CPU-baseline sweep)."""
    o = _Opts(opt.method, opt.max_iters, opt.fixed_pose, opt.jacobi_scaling, opt.phi, opt.huber_delta, opt.ftol,
              opt.gtol, opt.ptol, opt.radius0, opt.max_radius, opt.min_radius, opt.min_relative_decrease,
              opt.min_lm_diagonal, opt.max_lm_diagonal, opt.pcg_rtol, opt.pcg_max_iters, opt.threads, opt.verbose,
              opt.pcg_block_poses, opt.pcg_chain_len, 0)
    cap = opt.max_iters + 2
    recs = (_Iter * cap)()
    nrec = C.c_int(0)
    summ = _Summary()
    x = np.array(g.poses, np.float64, copy=True)
    ia, ib = np.ascontiguousarray(g.ia, np.int32), np.ascontiguousarray(g.ib, np.int32)
    meas, kind = np.ascontiguousarray(g.meas, np.float64), np.ascontiguousarray(g.kind, np.uint8)
    w = _info(g, opt.info_weighting)
    lib(variant).pgo_oracle_lm_pcg_w(g.n_poses, _dp(x), g.n_edges, _ip(ia), _ip(ib), _dp(meas), _dp(w), _bp(kind), C.byref(o), recs,
                              cap, C.byref(nrec), C.byref(summ))
    out = Result(x, summ.termination, summ.iterations, summ.successful_steps, summ.initial_cost, summ.final_cost)
    out.total_pcg_iters = summ.total_pcg_iters
    out.records = [{f[0]: getattr(recs[i], f[0]) for f in _Iter._fields_ if f[0] != "_pad"} for i in range(nrec.value)]
    out.seconds = dict(total=summ.seconds_total, eval=summ.seconds_eval, assemble=summ.seconds_assemble,
                       linear=summ.seconds_linear, candidate=summ.seconds_candidate)
    return out


def pcg_iterations(g: Graph, poses, poses0, radius: float, method: int = 1, rtol: float = 1e-10, block_poses: int = 32,
                   coarse_poses: int = 0, fixed_pose: int = 0, phi: float = 0.5, delta: float = 0.01, max_iters: int = 200000):
    """Restatement (numpy / scipy) of ONE linear solve of the PCG path at the LM state (poses, radius), for checking the
    preconditioners' iteration counts independently of the HIP kernels: the Jacobi-scaled system ((JS)'(JS) + D'D) y = S J'r
    with S from the Jacobian at `poses0` (Ceres: iteration 0), D'D = clamp(diag) / radius (identity on the constant pose);
    preconditioner = block-Jacobi over groups of `block_poses` consecutive poses (exact solves of the diagonal blocks) plus,
    for coarse_poses > 0, the additive coarse correction P (P'AP)^-1 P' on the rigid-body modes of aggregates of
    `coarse_poses` consecutive poses (translation x, y, rotation about the aggregate's centre; in scaled variables
    P_i = S_i^-1 B_i -- the second level of toy-robust-backend-slam_amd/csrc/coarse.hip.h, solved exactly here).
    Returns (PCG iterations to |r| <= rtol |b| from y = 0, solution y)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    N, E = g.n_poses, g.n_edges
    rows = np.repeat(np.arange(3 * E).reshape(E, 3), 6, axis=1).reshape(-1)
    cols = np.concatenate([3 * g.ia[:, None] + np.arange(3), 3 * g.ib[:, None] + np.arange(3)], axis=1)
    cols = np.tile(cols, (1, 3)).reshape(-1).astype(np.int64)

    def jac(p):
        _, r, J = evaluate(g, np.asarray(p, np.float64), method, phi, delta, True, True, True, 1)
        return sp.csr_matrix((J.reshape(-1), (rows, cols)), shape=(3 * E, 3 * N)), r.reshape(-1)

    A0, _ = jac(poses0)
    s = 1.0 / (1.0 + np.sqrt(np.asarray(A0.multiply(A0).sum(axis=0)).reshape(-1)))
    if fixed_pose >= 0:
        s[3 * fixed_pose:3 * fixed_pose + 3] = 0.0
    A, r = jac(poses)
    AS = A @ sp.diags(s)
    H = (AS.T @ AS).tocsr()
    d2 = np.clip(H.diagonal(), 1e-6, 1e32) / radius
    if fixed_pose >= 0:
        d2[3 * fixed_pose:3 * fixed_pose + 3] = 1.0
    H = (H + sp.diags(d2)).tocsr()
    b = AS.T @ r
    C = H.tocoo()
    grp = np.arange(3 * N) // (3 * block_poses)
    m = grp[C.row] == grp[C.col]
    M1 = spla.splu(sp.csc_matrix((C.data[m], (C.row[m], C.col[m])), shape=H.shape))
    apply_m = M1.solve
    if coarse_poses > 0:
        x = np.asarray(poses, np.float64).reshape(N, 3)
        agg = np.arange(N) // coarse_poses
        na = int(agg.max()) + 1
        cnt = np.bincount(agg, minlength=na)
        cx, cy = np.bincount(agg, x[:, 0], na) / cnt, np.bincount(agg, x[:, 1], na) / cnt
        in_graph = np.asarray(H.diagonal() - d2).reshape(N, 3).any(axis=1)       # poses without edges stay out
        sinv = np.where(s > 0, 1.0 / np.where(s > 0, s, 1.0), 0.0) * np.repeat(in_graph, 3)
        i = np.arange(N)
        pr = [3 * i, 3 * i + 1, 3 * i + 2, 3 * i, 3 * i + 1]
        pc = [3 * agg, 3 * agg + 1, 3 * agg + 2, 3 * agg + 2, 3 * agg + 2]
        pv = [sinv[3 * i], sinv[3 * i + 1], sinv[3 * i + 2], -(x[:, 1] - cy[agg]) * sinv[3 * i], (x[:, 0] - cx[agg]) * sinv[3 * i + 1]]
        P = sp.csr_matrix((np.concatenate(pv), (np.concatenate(pr), np.concatenate(pc))), shape=(3 * N, 3 * na))
        Ac = (P.T @ H @ P).toarray()
        dead = np.diag(Ac) == 0.0                      # an aggregate made of constant / edge-less poses only
        Ac[dead, dead] = 1.0
        Lc = np.linalg.cholesky(Ac)

        def apply_m(rv):
            rc = P.T @ rv
            ec = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rc))
            return M1.solve(rv) + P @ ec
    y = np.zeros_like(b)
    rv = b.copy()
    z = apply_m(rv)
    p = z.copy()
    rz = rv @ z
    bn = np.linalg.norm(b)
    for k in range(1, max_iters + 1):
        Ap = H @ p
        alpha = rz / (p @ Ap)
        y += alpha * p
        rv -= alpha * Ap
        if np.linalg.norm(rv) <= rtol * bn:
            return k, y
        z = apply_m(rv)
        rz2 = rv @ z
        p = z + (rz2 / rz) * p
        rz = rz2
    return max_iters, y


def lm_direct_sc(g: Graph, opt: Options = Options(), lam: float = 1.0) -> Result:
    """METHOD 2 (switchable constraints): the same Ceres policy as lm_direct on the joint parameter vector
    [poses; one switch per closure/bogus edge (initialised to 1.0)], residual blocks as in pgo_oracle_eval_sc,
    sparse direct solve.  Result.switches holds the final switch per edge (1.0 for odometry edges)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    N, E = g.n_poses, g.n_edges
    sc = np.nonzero(g.kind != 0)[0]
    M = len(sc)
    sw_col = -np.ones(E, np.int64)
    sw_col[sc] = 3 * N + np.arange(M)
    z = np.concatenate([np.array(g.poses, np.float64).reshape(-1), np.ones(M)])
    free = np.ones(3 * N + M, bool)
    if opt.fixed_pose >= 0:
        free[3 * opt.fixed_pose:3 * opt.fixed_pose + 3] = False
    free_idx = np.nonzero(free)[0]
    rows_p = np.repeat(np.arange(3 * E).reshape(E, 3), 6, axis=1).reshape(-1)
    cols_p = np.concatenate([3 * g.ia[:, None] + np.arange(3), 3 * g.ib[:, None] + np.arange(3)], axis=1)
    cols_p = np.tile(cols_p, (1, 3)).reshape(-1).astype(np.int64)
    rows_s = (3 * sc[:, None] + np.arange(3)).reshape(-1)
    cols_s = np.repeat(sw_col[sc], 3)
    rows_q = 3 * E + np.arange(M)
    sl = np.sqrt(lam)

    def ev(zv, with_j):
        sw = np.ones(E)
        sw[sc] = zv[3 * N:]
        cost, r, J, Js, q = evaluate_sc(g, zv[:3 * N].reshape(N, 3), sw, lam, opt.huber_delta, True, with_j, opt.threads)
        if not with_j:
            return cost, None, None
        A = sp.csr_matrix((np.concatenate([J.reshape(-1), Js[sc].reshape(-1), np.full(M, -sl)]),
                           (np.concatenate([rows_p, rows_s, rows_q]), np.concatenate([cols_p, cols_s, sw_col[sc]]))),
                          shape=(3 * E + M, 3 * N + M))
        return cost, np.concatenate([r.reshape(-1), q[sc]]), A[:, free_idx].tocsc()

    cost, rvec, A = ev(z, True)
    res = Result(z[:3 * N].reshape(N, 3), 4, 0, 0, cost, cost)
    if not np.isfinite(cost):
        res.termination = 6
        return res
    s = 1.0 / (1.0 + np.sqrt(np.asarray(A.multiply(A).sum(axis=0)).reshape(-1))) if opt.jacobi_scaling else np.ones(A.shape[1])
    grad = A.T @ rvec
    gmax = float(np.max(np.abs(grad)))
    x_norm = float(np.linalg.norm(z[free_idx]))
    radius, dec, prev_success, invalid_run = opt.radius0, 2.0, True, 0
    recs = [dict(iter=0, step_ok=1, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0,
                 relative_decrease=0.0, radius=radius)]
    it, term = 0, 4
    while True:
        it += 1
        if it > opt.max_iters:
            term, it = 4, it - 1
            break
        if prev_success and gmax <= opt.gtol:
            term, it = 2, it - 1
            break
        if radius < opt.min_radius:
            term, it = 5, it - 1
            break
        As = A @ sp.diags(s)
        H = (As.T @ As).tocsc()
        D2 = np.clip(H.diagonal(), opt.min_lm_diagonal, opt.max_lm_diagonal) / radius
        y = spla.splu((H + sp.diags(D2)).tocsc()).solve(s * grad)
        m = As @ (-y)
        model = float(-m @ (rvec + 0.5 * m))
        rec = dict(iter=it, step_ok=0, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0,
                   relative_decrease=0.0, radius=radius)
        if not np.all(np.isfinite(y)) or not (model > 0.0):
            invalid_run += 1
            if invalid_run >= 5:
                term = 6
                break
            radius /= dec
            dec *= 2.0
            prev_success = False
            rec.update(step_ok=-1, radius=radius)
            recs.append(rec)
            continue
        invalid_run = 0
        delta = np.zeros_like(z)
        delta[free_idx] = -s * y
        cand = z + delta
        cand_cost = ev(cand, False)[0]
        if not np.isfinite(cand_cost):
            cand_cost = np.finfo(np.float64).max
        step_norm = float(np.linalg.norm(delta))
        cost_change = cost - cand_cost
        rec.update(step_norm=step_norm, cost_change=cost_change)
        if step_norm <= opt.ptol * (x_norm + opt.ptol):
            term = 3
            recs.append(rec)
            break
        if abs(cost_change) <= opt.ftol * cost:
            term = 1
            recs.append(rec)
            break
        rho = cost_change / model if cand_cost < np.finfo(np.float64).max else -np.inf
        rec.update(relative_decrease=rho)
        if rho > opt.min_relative_decrease:
            z = cand
            x_norm = float(np.linalg.norm(z[free_idx]))
            cost, rvec, A = ev(z, True)
            if not np.isfinite(cost):
                term = 6
                break
            grad = A.T @ rvec
            gmax = float(np.max(np.abs(grad)))
            radius = min(opt.max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            dec, prev_success = 2.0, True
            res.successful_steps += 1
            rec.update(step_ok=1, cost=cost, gradient_max_norm=gmax)
        else:
            radius /= dec
            dec *= 2.0
            prev_success = False
            rec.update(step_ok=0, cost=cand_cost)
        rec.update(radius=radius)
        recs.append(rec)
    res.poses = z[:3 * N].reshape(N, 3)
    sw = np.ones(E)
    sw[sc] = z[3 * N:]
    res.switches = sw
    res.termination, res.iterations, res.final_cost, res.records = term, it, cost, recs
    return res
