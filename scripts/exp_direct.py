"""direct (chain + low-rank) solve vs PCG on the small datasets: GN it/s, final cost, distance to the golden direct-solve fixture"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
G = os.path.join(ROOT, "tests/golden")
cases = [("INTEL", 50, 1), ("INTEL", 50, 0), ("INTEL", 0, 1), ("MIT", 0, 1), ("MIT", 0, 0), ("CSAIL", 0, 1), ("FR079", 0, 1), ("MIT", 2, 1)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for name, n_out, method in cases:
    tag = "%s_out%d_m%d" % (name, n_out, method)
    ref = np.load(os.path.join(G, "lm_%s_poses.npy" % tag))
    fx = json.load(open(os.path.join(G, "lm_%s.json" % tag)))
    for ls in (2, 1):
        g = P.ReadG2O(os.path.join(G, "data/%s.g2o" % name))
        if n_out: g.add_random_C(n_out, 1)
        s = P.Solver(g, P.Options(method=method, linear_solver=ls, pcg_max_iters=400000))
        s.solve(); s.set_poses(np.array(g.poses))
        sm = s.solve()
        x = s.poses()
        i = s.info()
        recs = s.iter_records()
        hist = [r["step_ok"] for r in recs] == [r["step_ok"] for r in fx["records"]]
        relmax = max(r["pcg_rel_residual"] for r in recs)
        print("%-16s solver %d rank %4d: %7.1f GN it/s (%.2f ms/it, linear %.2f ms/it) it %d/%d cost %.12f (golden %.12f) dxy %.2e hist %s max rel res %.1e pcg %d" % (
            tag, i.linear_solver, i.direct_rank, sm.iterations / sm.seconds_total, 1e3 * sm.seconds_total / sm.iterations,
            1e3 * sm.seconds_linear / sm.iterations, sm.iterations, fx["iterations"], sm.final_cost, fx["final_cost"],
            np.abs(x[:, :2] - ref[:, :2]).max(), hist, relmax, sm.total_pcg_iters), flush=True)
        s.close()
