"""Replay one case of scripts/exp_shard_fuzz.py: shard_case.py WORLD 'RECIPE_JSON' 'OPTIONS_JSON' -- prints every rank's log on failure"""
import os, sys, json, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_sharded import run
world, recipe, opts = int(sys.argv[1]), json.loads(sys.argv[2]), json.loads(sys.argv[3])
tmp = tempfile.mkdtemp()
cfg = dict(graph="recipe", recipe=recipe, options=opts)
ref, rp = run(1, cfg, tmp, tag="r")
print("one rank:", ref[0]["summary"]["final_cost"], [r["pcg_iters"] for r in ref[0]["records"]], flush=True)
res, pp = run(world, cfg, tmp, tag="w")
print("%d ranks:" % world, res[0]["summary"]["final_cost"], [r["pcg_iters"] for r in res[0]["records"]], "d poses", float(np.abs(pp[0] - rp[0]).max()))
