"""dense 32-pose blocks vs chain-64 on the small real datasets (exact mode): time and PCG iterations of 50 LM iterations"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
for name, n_out, method in [("INTEL", 50, 1), ("INTEL", 50, 0), ("INTEL", 50, 2), ("M3500", 0, 1), ("M3500", 0, 0), ("M3500", 0, 2),
                            ("MIT", 0, 1), ("MIT", 0, 0), ("MIT", 0, 2), ("CSAIL", 0, 1), ("FR079", 0, 1), ("FRH", 0, 1)]:
    g = P.ReadG2O(os.path.join(ROOT, "tests", "golden", "data", name + ".g2o"))
    if n_out: g.add_random_C(n_out, 1)
    out = []
    for tag, kw in (("B32", dict(pcg_block_poses=32, pcg_chain_len=0)), ("chain64", dict(pcg_chain_len=64)), ("chain256", dict(pcg_chain_len=256))):
        s = P.Solver(g, P.Options(method=method, pcg_max_iters=400000, **kw))
        s.solve(); s.set_poses(np.array(g.poses))
        t = time.perf_counter(); summ = s.solve(); dt = time.perf_counter() - t
        out.append("%s: %.3f s %6d pcg" % (tag, dt, summ.total_pcg_iters))
        s.close()
    print("%-6s +%2d m%d  " % (name, n_out, method) + "   ".join(out), flush=True)
