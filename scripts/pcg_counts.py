import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import toy_robust_backend_slam_amd as P
for name, m in (("M3500", 1), ("M3500", 0), ("FRH", 1), ("FRH", 0)):
    g = P.ReadG2O("tests/golden/data/%s.g2o" % name)
    s = P.Solver(g, P.Options(method=m, linear_solver=1, pcg_max_iters=400000))
    sm = s.solve()
    print(name, m, [r["pcg_iters"] for r in s.iter_records()[1:]], "%.1f it/s" % (sm.iterations / sm.seconds_total))
    s.close()
