"""time to solution on the 1M-pose synthetic graph: a whole pgo_solve (Ceres defaults: 50 LM iterations max, ftol 1e-6) with
the bench's inexact PCG policy, chain vs dense preconditioner"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
g = P.synth_manhattan(1000000, 4.0, 0.10, 20260410)
x0 = np.array(g.poses)
for name, kw in (("chain-64", dict(pcg_chain_len=64)), ("dense B=4", dict(pcg_block_poses=4, pcg_chain_len=0))):
    s = P.Solver(g, P.Options(method=1, pcg_rtol=0.1, pcg_max_iters=500, **kw))
    s.solve(); s.set_poses(x0)          # warm-up (graph capture)
    t = time.perf_counter(); summ = s.solve(); dt = time.perf_counter() - t
    print(f"{name:10s}: {summ.iterations} LM iterations ({summ.successful_steps} accepted), termination {P.TERMINATION[summ.termination]}, "
          f"cost {summ.initial_cost:.3f} -> {summ.final_cost:.3f}, PCG iterations {summ.total_pcg_iters}, {dt:.3f} s "
          f"({summ.iterations/dt:.1f} GN it/s); linear {summ.seconds_linear:.3f} s eval {summ.seconds_eval:.3f} s assemble {summ.seconds_assemble:.3f} s", flush=True)
    s.close()
