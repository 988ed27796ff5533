"""small exact solves (INTEL + 50, MIT, M3500): chain length x apply kernel -> GN it/s, PCG iterations, us per PCG iteration"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import toy_robust_backend_slam_amd as P
    name, n_out, method, chain, block = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    g = P.ReadG2O(os.path.join(ROOT, "tests/golden/data/%s.g2o" % name))
    if n_out: g.add_random_C(n_out, 1)
    s = P.Solver(g, P.Options(method=method, pcg_chain_len=chain, pcg_block_poses=block, pcg_max_iters=400000))
    s.solve(); s.set_poses(np.array(g.poses))
    sm = s.solve()
    i = s.info()
    print("%-6s M%d chain %3d B %2d kernel %s: %6.1f GN it/s  pcg %6d  %5.2f us/pcg-it  final cost %.9f" % (
        name, method, i.pcg_chain_len, i.pcg_block_poses, os.environ.get("PGO_CHAIN_KERNEL", "auto"), sm.iterations / sm.seconds_total,
        sm.total_pcg_iters, 1e6 * sm.seconds_linear / max(1, sm.total_pcg_iters), sm.final_cost), flush=True)
    sys.exit(0)
cases = [("INTEL", 50, 1), ("MIT", 0, 1), ("M3500", 0, 1)]
for name, n_out, method in cases:
    for chain, block, kern in [(256, 0, "scan"), (256, 0, "lean4"), (128, 0, "lean2"), (128, 0, "lean4"), (64, 0, "lean2"), (64, 0, "lean4"), (32, 0, "lean2"), (0, 32, "auto"), (0, 16, "auto")]:
        env = dict(os.environ, PGO_CHAIN_KERNEL=kern)
        subprocess.run([sys.executable, __file__, "child", name, str(n_out), str(method), str(chain), str(block)], env=env, timeout=300)
