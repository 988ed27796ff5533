"""Round 3 experiment (GPU box): the second preconditioner level (rigid-body coarse modes, csrc/coarse.hip.h) against the
one-level solve: PCG iterations, GN it/s, distance to the golden direct-solve fixtures where they exist.
usage: exp_coarse.py [case ...]   cases: M3500 FRH INTEL s10k s100k s1m"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P

DATA = os.path.join(ROOT, "tests", "golden", "data")
GOLD = os.path.join(ROOT, "tests", "golden")

def run(tag, g, ref=None, budget=None, **kw):
    t0 = time.perf_counter()
    s = P.Solver(g, P.Options(**kw))
    t_create = time.perf_counter() - t0
    x0 = np.array(g.poses)
    if budget is None:
        s.solve(); s.set_poses(x0)
        sm = s.solve()
        its, secs, pcg, cost = sm.iterations, sm.seconds_total, sm.total_pcg_iters, sm.final_cost
    else:
        s.lm_begin(); s.lm_step(1); s.set_poses(x0); s.lm_begin()
        t0 = time.perf_counter(); its = 0; done = False
        while not done and time.perf_counter() - t0 < budget:
            done, sm = s.lm_step(1); its += 1
        secs, pcg, cost = time.perf_counter() - t0, sm.total_pcg_iters, sm.final_cost
    i = s.info()
    d = float(np.abs(s.poses()[:, :2] - ref[:, :2]).max()) if ref is not None else float("nan")
    print("%-44s coarse %4d (rank %4d) solver %d: %4d LM it, %8d PCG it, %8.3f s -> %7.1f GN it/s, cost %.6f, |dxy| vs golden %.2e, create %.2f s"
          % (tag, i.pcg_coarse_poses, i.pcg_coarse_rank, i.linear_solver, its, pcg, secs, its / secs, cost, d, t_create), flush=True)
    s.close()

cases = sys.argv[1:] or ["M3500", "FRH", "s10k"]
for c in cases:
    if c in ("M3500", "FRH", "INTEL", "MIT"):
        g = P.ReadG2O(os.path.join(DATA, c + ".g2o"))
        for m in (1, 0):
            ref = np.load(os.path.join(GOLD, "lm_%s_out0_m%d_poses.npy" % (c, m))) if os.path.exists(os.path.join(GOLD, "lm_%s_out0_m%d_poses.npy" % (c, m))) else None
            run("%s METHOD %d one level PCG" % (c, m), g, ref, method=m, linear_solver=1, pcg_coarse_poses=0, pcg_max_iters=20000)
            for a in (-1, 8, 16, 32, 64):
                run("%s METHOD %d two levels (%d)" % (c, m, a), g, ref, method=m, linear_solver=1, pcg_coarse_poses=a, pcg_max_iters=20000)
            run("%s METHOD %d library default" % (c, m), g, ref, method=m, pcg_max_iters=20000)
    else:
        n = {"s10k": 10000, "s100k": 100000, "s1m": 1000000}[c]
        g = P.synth_manhattan(n, 4.0, 0.10, 20260410)
        base = dict(method=1, max_iters=100000, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0, pcg_max_iters=60000)
        for rtol in ((1e-10, 1e-6, 1e-3, 0.1) if n <= 100000 else (1e-3, 0.1)):
            for a in ((0, -1, 16, 64, 256) if n == 10000 else (0, 64, 128, 256, 512) if n == 100000 else (0, 512, 1024, 2048)):
                run("synthetic %s rtol %g agg %d" % (c, rtol, a), g, None, budget=2.0, pcg_rtol=rtol, pcg_coarse_poses=a,
                    pcg_check_every=10 if rtol >= 0.1 else 50, **base)
