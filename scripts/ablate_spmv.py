import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import toy_robust_backend_slam_amd as P
g = P.synth_manhattan(1000000)
s = P.Solver(g, P.Options(method=1, max_iters=2, pcg_rtol=0.1, pcg_max_iters=50))
s.lm_begin(); s.lm_step(1)
for mode in (0, 1, 2, 3, 0):
    os.environ["PGO_SPMV_ABLATE"] = str(mode)
    k = s.bench_spmv(20)
    print("mode", mode, "ms", round(k.ms_avg, 4))
