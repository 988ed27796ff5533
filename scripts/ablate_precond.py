"""time the preconditioner apply (PCG start-up kernel) for the chain / dense-block / 3x3 variants, with the chain
scan ablations (PGO_CHAIN_ABLATE: 1 = no backward scan, 2 = no scans; timing only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import toy_robust_backend_slam_amd as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = P.synth_manhattan(N, 4.0, 0.10, 20260410)
for name, kw in [("chain-64", dict(pcg_chain_len=64)), ("dense B=4", dict(pcg_block_poses=4, pcg_chain_len=0)), ("3x3", dict(pcg_block_poses=1, pcg_chain_len=0))]:
    s = P.Solver(g, P.Options(method=1, max_iters=2, ftol=0.0, gtol=0.0, ptol=0.0, pcg_rtol=0.1, pcg_max_iters=500, **kw))
    s.lm_begin(); s.lm_step(2)
    for abl in ((0, 1, 2) if name == "chain-64" else (0,)):
        os.environ["PGO_CHAIN_ABLATE"] = str(abl)
        k = s.bench_precond(20)
        print(f"{name:10s} ablate {abl}: {k.ms_avg*1e3:7.1f} us  {k.algorithmic_bytes/(k.ms_avg*1e-3)/1e9:7.0f} GB/s algorithmic", flush=True)
    s.close()
