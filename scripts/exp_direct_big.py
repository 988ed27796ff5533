import json, os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd(); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
G = os.path.join(ROOT, "tests/golden")
for name, n_out, method in [("M3500", 0, 1), ("M3500", 0, 0), ("FRH", 0, 1), ("M3500", 184, 1)]:
    tag = "%s_out%d_m%d" % (name, n_out, method)
    ref = np.load(os.path.join(G, "lm_%s_poses.npy" % tag)); fx = json.load(open(os.path.join(G, "lm_%s.json" % tag)))
    for ls in (2, 1):
        g = P.ReadG2O(os.path.join(G, "data/%s.g2o" % name))
        if n_out: g.add_random_C(n_out, 1)
        s = P.Solver(g, P.Options(method=method, linear_solver=ls, pcg_max_iters=400000))
        s.solve(); s.set_poses(np.array(g.poses)); sm = s.solve(); x = s.poses(); i = s.info(); recs = s.iter_records()
        print("%-16s solver %d rank %4d: %7.1f GN it/s (%.2f ms/it) it %d/%d cost %.12f (golden %.12f) dxy %.2e hist %s max rel res %.1e pcg %d" % (
            tag, i.linear_solver, i.direct_rank, sm.iterations / sm.seconds_total, 1e3 * sm.seconds_total / sm.iterations, sm.iterations, fx["iterations"], sm.final_cost, fx["final_cost"],
            np.abs(x[:, :2] - ref[:, :2]).max(), [r["step_ok"] for r in recs] == [r["step_ok"] for r in fx["records"]], max(r["pcg_rel_residual"] for r in recs), sm.total_pcg_iters), flush=True)
        s.close()
