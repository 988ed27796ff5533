"""one small exact solve: python scripts/small_child.py NAME N_OUT METHOD CHAIN  (env: PGO_SOLO, PGO_CHAIN_KERNEL, PGO_CHAIN_SCAN)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import toy_robust_backend_slam_amd as P
name, n_out, method, chain = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = P.ReadG2O(os.path.join(ROOT, "tests/golden/data/%s.g2o" % name))
if n_out: g.add_random_C(n_out, 1)
s = P.Solver(g, P.Options(method=method, pcg_chain_len=chain, pcg_max_iters=400000))
s.solve(); s.set_poses(np.array(g.poses))
sm = s.solve()
i = s.info()
ref = None
try:
    ref = np.load(os.path.join(ROOT, "tests/golden/lm_%s_out%d_m%d_poses.npy" % (name, n_out, method)))
except Exception: pass
d = float(np.abs(s.poses()[:, :2] - ref[:, :2]).max()) if ref is not None else float("nan")
print("%-6s M%d chain %3d kernel %d solo=%s scan=%s: %6.1f GN it/s  pcg %6d  %5.2f us/pcg-it  cost %.9f  max dxy vs golden %.2e" % (
    name, method, i.pcg_chain_len, i.chain_kernel, os.environ.get("PGO_SOLO", "-"), os.environ.get("PGO_CHAIN_SCAN", "-"), sm.iterations / sm.seconds_total, sm.total_pcg_iters,
    1e6 * sm.seconds_linear / max(1, sm.total_pcg_iters), sm.final_cost, d), flush=True)
