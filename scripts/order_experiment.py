"""Experiment (not product code): how much would an internal locality-aware pose ordering buy on ONE GPU?
The graph is permuted in Python before the solver is created; measured: k_spmv / k_assemble / k_edge_eval time and
PCG iterations + GN it/s of 6 LM iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
import toy_robust_backend_slam_amd as P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = P.synth_manhattan(N, 4.0, 0.10, 20260410)
poses, ia, ib, meas, kind = (np.array(x) for x in (g.poses, g.ia, g.ib, g.meas, g.kind))
ia64, ib64 = ia.astype(np.int64), ib.astype(np.int64)
E = len(ia)

def supported():
    key = np.minimum(ia64, ib64) * N + np.maximum(ia64, ib64)
    keys = np.sort(key)
    def has(x, y):
        ok = (x >= 0) & (y >= 0) & (x < N) & (y < N) & (x != y)
        k = np.minimum(x, y) * N + np.maximum(x, y)
        pos = np.searchsorted(keys, k)
        pos[pos >= len(keys)] = len(keys) - 1
        return ok & (keys[pos] == k)
    sup = np.abs(ia64 - ib64) <= 1
    nc = ~sup
    acc = np.zeros(E, bool)
    for da in (-1, 0, 1):
        for db in (-1, 0, 1):
            if da == 0 and db == 0: continue
            acc |= has(ia64 + da, ib64 + db)
    return sup | (nc & acc)

def run(tag, perm):
    # perm: old index -> new index.  pose 0 must stay the constant pose: fixed_pose = perm[0]
    inv = np.empty(N, np.int64); inv[perm] = np.arange(N)
    g2 = P.Graph.from_arrays(poses[inv], perm[ia].astype(np.int32), perm[ib].astype(np.int32), meas, kind)
    s = P.Solver(g2, P.Options(method=1, max_iters=6, ftol=0.0, gtol=0.0, ptol=0.0, pcg_rtol=0.1, pcg_max_iters=500, fixed_pose=int(perm[0])))
    s.solve(); s.set_poses(poses[inv])
    t = time.perf_counter(); summ = s.solve(); dt = time.perf_counter() - t
    k3 = s.bench_spmv(20); k2 = s.bench_assemble(10); k1 = s.bench_eval(10, True)
    print(f"{tag:28s} GN it/s {summ.iterations/dt:6.1f}  pcg {summ.total_pcg_iters:5d}  cost {summ.final_cost:.3f}  spmv {k3.ms_avg*1e3:6.1f} us  asm {k2.ms_avg*1e3:6.1f} us  eval {k1.ms_avg*1e3:6.1f} us", flush=True)
    s.close()

run("natural", np.arange(N, dtype=np.int64))
t = time.time(); sup = supported(); print("support filter", round(time.time() - t, 1), "s; kept closures", int(sup[kind == 1].sum()), "of", int((kind == 1).sum()), "bogus", int(sup[kind == 2].sum()), "of", int((kind == 2).sum()))
for L in (64, 256, 1024):
    ns = (N + L - 1) // L
    sa, sb = ia64[sup] // L, ib64[sup] // L
    m = sa != sb
    A = sp.coo_matrix((np.ones(m.sum()), (sa[m], sb[m])), shape=(ns, ns)); A = (A + A.T).tocsr()
    order = reverse_cuthill_mckee(A, symmetric_mode=True)        # order[k] = old segment at new position k
    pos = np.empty(ns, np.int64); pos[order] = np.arange(ns)
    # new index of pose i: segments keep their internal order; the last (short) segment must stay last
    seg_len = np.full(ns, L, np.int64); seg_len[ns - 1] = N - L * (ns - 1)
    start = np.zeros(ns, np.int64); start[order] = np.concatenate([[0], np.cumsum(seg_len[order])[:-1]])
    perm = start[np.arange(N) // L] + (np.arange(N) % L)
    assert len(np.unique(perm)) == N
    run("segment RCM, L=%d" % L, perm)
