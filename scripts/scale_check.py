"""4 ranks (shm back-end, one GPU) vs 1 rank on a 500k-pose synthetic graph with the bench options (chain preconditioner,
internal pose ordering, overlapped halo exchange): LM history and poses must agree.  Run on the GPU box."""
import sys, os, json, subprocess, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,'tests'))
os.chdir(ROOT)
WORKER=os.path.join(ROOT,'tests','_shard_worker.py')
def run(world,cfg,out):
    os.makedirs(out,exist_ok=True)
    name="pgo_scale_%d_%d"%(os.getpid(),world)
    procs=[subprocess.Popen([sys.executable,WORKER,json.dumps(dict(cfg,rank=r,world=world,name=name,out=out))],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True) for r in range(world)]
    for p in procs:
        o,_=p.communicate(timeout=900)
        assert p.returncode==0,o
    return [json.load(open(os.path.join(out,"out_%d.json"%r))) for r in range(world)],[np.load(os.path.join(out,"poses_%d.npy"%r)) for r in range(world)]
tmp=tempfile.mkdtemp()
cfg=dict(graph="synth",n_poses=500000,seed=20260410,options=dict(method=1,max_iters=5,pcg_rtol=0.1,pcg_max_iters=500,halo_exchange=1))
ref,rp=run(1,cfg,tmp+"/w1")
res,pp=run(4,cfg,tmp+"/w4")
for a,b in zip(res[0]["records"],ref[0]["records"]):
    print("iter",a["iter"],"cost rel diff %.2e"%(abs(a["cost"]-b["cost"])/b["cost"]),"pcg",a["pcg_iters"],b["pcg_iters"],a["step_ok"],b["step_ok"])
print("max pose diff vs 1 rank", np.abs(pp[0]-rp[0]).max(), "ranks identical", all(np.array_equal(pp[r],pp[0]) for r in range(4)))
