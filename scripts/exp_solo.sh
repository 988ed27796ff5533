#!/bin/bash
set -o pipefail
python - <<'PY'
import os, sys, subprocess
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import toy_robust_backend_slam_amd as P
name, n_out, method, chain = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = P.ReadG2O("tests/golden/data/%s.g2o" % name)
if n_out: g.add_random_C(n_out, 1)
s = P.Solver(g, P.Options(method=method, pcg_chain_len=chain, pcg_max_iters=400000))
s.solve(); s.set_poses(np.array(g.poses))
sm = s.solve()
i = s.info()
ref = None
try:
    ref = np.load("tests/golden/lm_%s_out%d_m%d_poses.npy" % (name, n_out, method))
except Exception: pass
d = float(np.abs(s.poses()[:, :2] - ref[:, :2]).max()) if ref is not None else float("nan")
print("%-6s M%d chain %3d solo=%s: %6.1f GN it/s  pcg %6d  %5.2f us/pcg-it  cost %.9f  max dxy vs golden %.2e" % (
    name, method, i.pcg_chain_len, os.environ.get("PGO_SOLO", "1"), sm.iterations / sm.seconds_total, sm.total_pcg_iters,
    1e6 * sm.seconds_linear / max(1, sm.total_pcg_iters), sm.final_cost, d), flush=True)
'''
open("/tmp/solo_child.py", "w").write(code)
for name, n_out, method in [("INTEL", 50, 1), ("INTEL", 50, 0), ("MIT", 0, 1), ("CSAIL", 0, 1), ("FR079", 0, 1)]:
    for chain in (256, 64, -1):
        for solo in ("0", "1"):
            subprocess.run([sys.executable, "/tmp/solo_child.py", name, str(n_out), str(method), str(chain)], env=dict(os.environ, PGO_SOLO=solo), timeout=300)
PY
