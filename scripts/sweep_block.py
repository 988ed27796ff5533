import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
for n in (1000000,):
    g = P.synth_manhattan(n, 4.0, 0.10, 20260410)
    for B in (3, 4, 5, 6):
        s = P.Solver(g, P.Options(method=1, max_iters=10, ftol=0.0, gtol=0.0, ptol=0.0, pcg_rtol=0.1, pcg_max_iters=500, pcg_check_every=50, pcg_block_poses=B))
        s.solve(); s.set_poses(np.array(g.poses))
        t = time.perf_counter(); summ = s.solve(); dt = time.perf_counter() - t
        print(f"n={n} B={B}: {summ.iterations/dt:.1f} it/s  pcg {summ.total_pcg_iters}  cost {summ.final_cost:.4f}  lin {summ.seconds_linear:.3f}s", flush=True)
        s.close()
