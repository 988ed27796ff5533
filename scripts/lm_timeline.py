import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
g = P.synth_manhattan(1000000, 4.0, 0.10, 20260410)
for every in (100, 25):
    s = P.Solver(g, P.Options(method=1, max_iters=8, ftol=0.0, gtol=0.0, ptol=0.0, pcg_rtol=0.1, pcg_max_iters=500, pcg_check_every=every))
    s.solve(); s.set_poses(np.array(g.poses))
    t = time.perf_counter(); summ = s.solve(); dt = time.perf_counter() - t
    print("check_every", every, "total", round(dt*1e3,1), "ms", {k: round(v*1e3,1) for k, v in summ.as_dict().items() if k.startswith("seconds")})
    for r in s.iter_records(): print("  it", r["iter"], "pcg", r["pcg_iters"], "ms", round(r["seconds"]*1e3, 2), "per-pcg us", round(r["seconds"]*1e6/max(1,r["pcg_iters"]),1))
    s.close()
