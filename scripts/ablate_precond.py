"""time the preconditioner apply (PCG start-up kernel) for the chain / dense-block / 3x3 variants"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import toy_robust_backend_slam_amd as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = P.synth_manhattan(N, 4.0, 0.10, 20260410)
for name, kw in [("chain-64", dict(pcg_chain_len=64)), ("chain-256", dict(pcg_chain_len=256)), ("dense B=4", dict(pcg_block_poses=4, pcg_chain_len=0)), ("3x3", dict(pcg_block_poses=1, pcg_chain_len=0))]:
    s = P.Solver(g, P.Options(method=1, max_iters=8, ftol=0.0, gtol=0.0, ptol=0.0, pcg_rtol=0.1, pcg_max_iters=500, **kw))
    s.lm_begin(); s.lm_step(2)
    k = s.bench_precond(20)
    summ = s.lm_step(6)[1]
    print(f"{name:10s}: {k.ms_avg*1e3:7.1f} us  {k.algorithmic_bytes/(k.ms_avg*1e-3)/1e9:7.0f} GB/s algorithmic; PCG iterations over 8 LM iterations {summ.total_pcg_iters}", flush=True)
    s.close()
