"""mid-size synthetic graphs (inexact mode): GN it/s for a few settings (env passed through)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
for n in (60000, 100000, 200000, 300000):
    g = P.synth_manhattan(n, 4.0, 0.10, 20260410)
    s = P.Solver(g, P.Options(method=1, max_iters=50, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0, pcg_rtol=0.1, pcg_max_iters=500, pcg_check_every=10))
    x0 = np.array(g.poses)
    s.solve(); best = None
    for _ in range(3):
        s.set_poses(x0); sm = s.solve()
        if best is None or sm.seconds_total < best.seconds_total: best = sm
    i = s.info()
    print("N %7d: %6.1f GN it/s  %5.1f us per PCG it  (pcg %d, chain %d, B %d, kernel %d)  env %s" % (
        n, best.iterations / best.seconds_total, 1e6 * best.seconds_linear / best.total_pcg_iters, best.total_pcg_iters, i.pcg_chain_len, i.pcg_block_poses, i.chain_kernel,
        {k: v for k, v in os.environ.items() if k.startswith("PGO_")}), flush=True)
    s.close()
