"""phase stamps inside k_cg_update1_cl on a small graph (experiment build libpgo_phase.so, -DPGO_PHASE_TIMING)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["PGO_LIB"] = os.path.join(ROOT, "toy-robust-backend-slam_amd", "libpgo_phase.so")
import numpy as np
import toy_robust_backend_slam_amd as P
g = P.ReadG2O(os.path.join(ROOT, "tests/golden/data/INTEL.g2o")); g.add_random_C(50, 1)
for ug in (1, 0):
    s = P.Solver(g, P.Options(method=1, max_iters=3, use_graphs=ug))
    s.solve()
    t = (C.c_ulonglong * 16)()
    P.lib().pgo_debug_phase_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    P.lib().pgo_debug_phase_times(s._h, t)
    v = [t[i] for i in range(7)]
    print("use_graphs", ug, "chain kernel", s.info().chain_kernel, "phase deltas (us, 100 MHz clock):", [round((v[i + 1] - v[i]) / 100.0, 2) for i in range(6)],
          "= start->done-flag, ->alpha, ->vectors+LDS, ->apply, ->z store, ->reductions")
    s.close()
