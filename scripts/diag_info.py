"""diagnostic: per-iteration cost difference GPU (PCG) vs golden (direct) in the information-weighted mode"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
G = os.path.join(ROOT, "tests", "golden")
for name, n_out, method in [("M3500", 0, 1), ("INTEL", 50, 1)]:
    tag = "%s_out%d_m%d_info" % (name, n_out, method)
    fx = json.load(open(os.path.join(G, "lm_%s.json" % tag)))
    ref = np.load(os.path.join(G, "lm_%s_poses.npy" % tag))
    for rtol in (1e-10, 1e-13):
        g = P.ReadG2O(os.path.join(G, "data", name + ".g2o"))
        if n_out: g.add_random_C(n_out, 1)
        s = P.Solver(g, P.Options(method=method, info_weighting=1, phi=1.0, pcg_max_iters=2000000, pcg_rtol=rtol))
        summ = s.solve(); recs = s.iter_records()
        d = [abs(a["cost"] - b["cost"]) / b["cost"] for a, b in zip(recs, fx["records"])]
        print(tag, rtol, "pcg", summ.total_pcg_iters, "dxy", np.abs(s.poses()[:, :2] - ref[:, :2]).max())
        print("  rel cost diff by iter:", " ".join("%.0e" % v for v in d))
        print("  pcg rel res:", " ".join("%.0e" % r["pcg_rel_residual"] for r in recs[1:]))
        print("  radius:", " ".join("%.0e" % r["radius"] for r in recs[1:]))
        s.close()
