"""Probe of the hot kernels at the bench workload for rocprofv3 --pmc passes and quick A/B timings (run on the GPU box).
usage: k3_probe.py [pose_ordering] [poses] [spmv_pipe knob]      prints HIP-event times of K3 / K2 / K1 / the preconditioner apply"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toy_robust_backend_slam_amd as P

po = int(sys.argv[1]) if len(sys.argv) > 1 else -1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
depth = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # test hook "spmv_pipe": 2 = the two-deep pipelined K3
if depth >= 0:
    P.set_knob("spmv_pipe", depth)
g = P.synth_manhattan(n, 4.0, 0.10, 20260410)
s = P.Solver(g, P.Options(method=1, max_iters=2, pcg_rtol=0.1, pcg_max_iters=50, pose_ordering=po))
s.lm_begin(); s.lm_step(1)
k3 = s.bench_spmv(6); k2 = s.bench_assemble(3); k1 = s.bench_eval(3, True); k1c = s.bench_eval(3, False); pc = s.bench_precond(3)
x = __import__("numpy").random.default_rng(1).standard_normal(3 * g.n_poses)
chk = float(abs(s.spmv(x)).sum())
print("spmv_pipe %d checksum %.12e | pose_ordering %d (resolved %d): k_spmv %.1f us  k_assemble %.1f us  k_edge_eval<jac> %.1f us  <cost> %.1f us  precond %.1f us" % (
    depth, chk, po, s.info().pose_ordering, k3.ms_avg * 1e3, k2.ms_avg * 1e3, k1.ms_avg * 1e3, k1c.ms_avg * 1e3, pc.ms_avg * 1e3), flush=True)
s.close()
