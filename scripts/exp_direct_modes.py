"""direct solve in the other modes (METHOD 2 switchable constraints, information weighting) vs PCG and the golden fixtures"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
G = os.path.join(ROOT, "tests/golden")
for name, n_out, kw, tag in [("INTEL", 50, dict(method=2), "INTEL_out50_m2"), ("MIT", 0, dict(method=2), "MIT_out0_m2"),
                             ("INTEL", 50, dict(method=1, info_weighting=1), "INTEL_out50_m1_info"), ("MIT", 0, dict(method=0, info_weighting=1), "MIT_out0_m0_info")]:
    if not os.path.exists(os.path.join(G, "lm_%s.json" % tag)):
        print("no fixture", tag); continue
    fx = json.load(open(os.path.join(G, "lm_%s.json" % tag))); ref = np.load(os.path.join(G, "lm_%s_poses.npy" % tag))
    if "phi" in fx: kw = dict(kw, phi=fx["phi"])
    for ls in (2, 1):
        g = P.ReadG2O(os.path.join(G, "data/%s.g2o" % name))
        if n_out: g.add_random_C(n_out, 1)
        s = P.Solver(g, P.Options(linear_solver=ls, pcg_max_iters=2000000, pcg_rtol=1e-13 if "info_weighting" in kw else 1e-10, **kw))
        s.solve(); s.set_poses(np.array(g.poses)); sm = s.solve(); x = s.poses(); recs = s.iter_records()
        hist = [r["step_ok"] for r in recs] == [r["step_ok"] for r in fx["records"]]
        print("%-22s solver %d rank %4d: %7.1f GN it/s it %d/%d cost %.12f (golden %.12f) dxy %.2e hist %s max rel res %.1e pcg %d" % (
            tag, s.info().linear_solver, s.info().direct_rank, sm.iterations / sm.seconds_total, sm.iterations, fx["iterations"], sm.final_cost, fx["final_cost"],
            np.abs(x[:, :2] - ref[:, :2]).max(), hist, max(r["pcg_rel_residual"] for r in recs), sm.total_pcg_iters), flush=True)
        s.close()
