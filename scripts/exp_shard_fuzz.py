"""Sharded solves (shm communicator, 2-4 ranks on one GPU) of arbitrary small graphs against the one-rank solve: shards that
own nothing, graphs smaller than the alignment, several components, hubs.  Run on the GPU box; prints one line per case."""
import os, sys, json, subprocess, tempfile, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_sharded import run
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for case in range(n_cases):
    n = int(rng.choice([3, 9, 63, 64, 65, 130, 257, 500, 1500]))
    recipe = [n, int(rng.integers(1 << 30)), float(rng.choice([1.0, 1.0, 0.6])), int(rng.integers(0, 2 * n)), bool(rng.integers(2)) and n > 8, int(rng.integers(0, 3))]
    world = int(rng.choice([2, 3, 4]))
    opts = dict(method=int(rng.choice([0, 1, 2])), fixed_pose=int(rng.choice([0, -1])), max_iters=3, pcg_rtol=float(rng.choice([1e-12, 1e-3])), pcg_max_iters=100000,
                linear_solver=1, halo_exchange=int(rng.choice([0, 1])), pcg_chain_len=int(rng.choice([-1, 8, 64, 0])))
    knobs = dict(shm_timeout_s=15)
    if rng.integers(2):
        knobs["pad_tiles"] = 1        # the large-graph layout (padded tile slots, k_spmv_1, folded partials) on small shards
    cfg = dict(graph="recipe", recipe=recipe, options=opts, knobs=knobs)
    tag = "c%d" % case
    try:
        ref, rp = run(1, cfg, tmp, tag=tag + "r")
        res, pp = run(world, cfg, tmp, tag=tag)
        ok = all(np.array_equal(pp[r], pp[0]) for r in range(world))
        hist = [a["step_ok"] for a in res[0]["records"]] == [b["step_ok"] for b in ref[0]["records"]]
        c0 = max(ref[0]["summary"]["initial_cost"], 1e-30)
        dc = abs(res[0]["summary"]["final_cost"] - ref[0]["summary"]["final_cost"]) / c0
        dp = float(np.abs(pp[0] - rp[0]).max())
        tight = opts["pcg_rtol"] < 1e-6
        good = ok and hist and (not tight or (dc < 1e-8 and dp < 1e-6 * max(1.0, np.abs(rp[0]).max())))
        print("case %2d n=%4d world=%d %s pad %s: ranks identical %s, history %s, d cost %.1e, d poses %.1e, pcg %d vs %d %s" % (
            case, n, world, json.dumps(opts), knobs.get("pad_tiles", -1), ok, hist, dc, dp, res[0]["summary"]["total_pcg_iters"], ref[0]["summary"]["total_pcg_iters"], "" if good else "  <-- CHECK"), flush=True)
        bad += not good
    except AssertionError as e:
        print("case %2d n=%4d world=%d recipe %r %s: FAILED\n%s" % (case, n, world, recipe, json.dumps(opts), str(e)[-1500:]), flush=True)
        bad += 1
print("cases to check:", bad)
