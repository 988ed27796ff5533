"""Replay one recipe of tests/test_gpu_fuzz.py (the line a failing example prints) through the PCG loops at several tolerances.
usage: fuzz_case.py n seed p_chain n_extra hub dup fixed method"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import toy_robust_backend_slam_amd as P
from test_gpu_fuzz import make_graph
n, seed, p_chain, n_extra, hub, dup, fixed, method = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "True", int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
poses, ia, ib, meas, kind = make_graph(n, seed, p_chain, n_extra, hub, dup)
g = P.Graph.from_arrays(poses, ia, ib, meas, kind)
def run(label, rtol, sr, iters=3):
    kw = dict(method=method, fixed_pose=fixed, max_iters=iters, pcg_rtol=rtol, pcg_max_iters=200000, linear_solver=1, pcg_chain_len=8, pcg_coarse_poses=0)
    P.set_knob("single_reduction", 1 if sr else 0); P.set_knob("fused_p", 0)
    try:
        s = P.Solver(g, P.Options(**kw))
    finally:
        P.set_knob("single_reduction", -1); P.set_knob("fused_p", -1)
    sm = s.solve()
    print("%-28s final cost %.12f  " % (label, sm.final_cost), [(r["pcg_iters"], "%.1e" % r["pcg_rel_residual"], r["step_ok"], "%.6f" % r["cost"]) for r in s.iter_records()])
    x = s.poses().copy(); s.close()
    return x
ref = run("two reductions, 1e-13", 1e-13, False)
for rtol in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
    a = run("two reductions, %g" % rtol, rtol, False)
    b = run("one reduction,  %g" % rtol, rtol, True)
    print("   distance to the tight solve: two %.2e  one %.2e" % (np.abs(a - ref).max(), np.abs(b - ref).max()))
