"""GN (LM) iterations/s on every workload north_star names, next to the CPU oracle: INTEL (+50 outliers), M3500, MIT
with the reference's exact-solve semantics (PCG to 1e-10 standing in for SPARSE_NORMAL_CHOLESKY; the oracle runs the
direct solve), and synthetic 10k / 100k / 1M with the inexact policy (eta 0.1, <= 500 PCG iterations; the oracle
runs the C port of the same algorithm).  Writes profiles/<tag>_workloads.json + .md.   Run on the GPU box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import toy_robust_backend_slam_amd as P  # noqa: E402


def og(g):
    return O.Graph(np.array(g.pose_ids), np.array(g.poses), np.array(g.ia), np.array(g.ib), np.array(g.meas),
                   np.array(g.info), np.array(g.kind))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    threads = min(16, os.cpu_count() or 1)
    rows = []
    for name, n_out, method in [("INTEL", 50, 1), ("INTEL", 50, 0), ("INTEL", 50, 2), ("M3500", 0, 1), ("M3500", 0, 0), ("M3500", 0, 2),
                                ("MIT", 0, 1), ("MIT", 0, 0), ("MIT", 0, 2)]:
        g = P.ReadG2O(os.path.join(ROOT, "tests", "golden", "data", name + ".g2o"))
        if n_out:
            g.add_random_C(n_out, 1)
        d = np.abs(np.array(g.ia).astype(np.int64) - np.array(g.ib).astype(np.int64))
        chain_like = 20 * int(((d >= 2) & (d < 32)).sum()) <= g.n_poses      # pgo_internal.h resolve_chain_len
        s = P.Solver(g, P.Options(method=method, pcg_max_iters=200000))
        s.solve()  # warm-up (graph capture, clocks)
        s.set_poses(np.array(g.poses))
        t = time.perf_counter()
        summ = s.solve()
        dt = time.perf_counter() - t
        x = s.poses()
        s.close()
        t = time.perf_counter()
        ores = O.lm_direct_sc(og(g), O.Options(method=2)) if method == 2 else O.lm_direct(og(g), O.Options(method=method))
        odt = time.perf_counter() - t
        rows.append(dict(workload="%s +%d outliers, METHOD %d" % (name, n_out, method), poses=g.n_poses, edges=g.n_edges,
                         policy="exact (PCG rtol 1e-10, %s)" % ("256-pose chain segments" if chain_like else "32-pose blocks"), lm_iters=summ.iterations, pcg_iters=summ.total_pcg_iters,
                         gpu_s=dt, gpu_it_s=summ.iterations / dt, cpu_kind="oracle lm_direct (scipy SuperLU + C eval, 1 thread)",
                         cpu_s=odt, cpu_it_s=ores.iterations / odt, final_cost_gpu=summ.final_cost, final_cost_cpu=ores.final_cost,
                         max_dxy=float(np.abs(x[:, :2] - ores.poses[:, :2]).max())))
        print(rows[-1], flush=True)
    for n in (10000, 100000, 1000000):
        g = P.synth_manhattan(n, 4.0, 0.10, 20260410)
        iters = 10
        chain = n > 50000   # the library's auto rule (pgo_internal.h resolve_chain_len)
        kw = dict(method=1, max_iters=iters, ftol=0.0, gtol=0.0, ptol=0.0, pcg_rtol=0.1, pcg_max_iters=500,
                  pcg_chain_len=64 if chain else 0, pcg_block_poses=0 if chain else 4)
        s = P.Solver(g, P.Options(pcg_check_every=100, **kw))
        s.solve()
        s.set_poses(np.array(g.poses))
        t = time.perf_counter()
        summ = s.solve()
        dt = time.perf_counter() - t
        x = s.poses()
        k1 = s.bench_eval(10, True)
        s.close()
        cpu_iters = iters if n <= 100000 else 3
        t = time.perf_counter()
        ores = O.lm_pcg(og(g), O.Options(threads=threads, **dict(kw, max_iters=cpu_iters)))
        odt = time.perf_counter() - t
        rows.append(dict(workload="synthetic Manhattan %d poses, 10%% outliers, DCS" % n, poses=g.n_poses, edges=g.n_edges,
                         policy="inexact (eta 0.1, <= 500 PCG, %s)" % ("64-pose chain segments" if chain else "4-pose blocks"), lm_iters=summ.iterations, pcg_iters=summ.total_pcg_iters,
                         gpu_s=dt, gpu_it_s=summ.iterations / dt, cpu_kind="oracle lm_pcg (C port, %d threads)" % threads,
                         cpu_s=odt, cpu_it_s=ores.iterations / odt, final_cost_gpu=summ.final_cost,
                         final_cost_cpu=ores.final_cost, edges_per_s_k1=k1.units / (k1.ms_avg * 1e-3),
                         max_dxy=float(np.abs(x[:, :2] - ores.poses[:, :2]).max()) if cpu_iters == iters else None))
        print(rows[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "profiles", tag + "_workloads.json"), "w"), indent=1)
    md = ["# GN (LM) iterations/s per workload, 1 x MI355X vs CPU oracle (%s)" % tag, "",
          "| workload | poses | edges | policy | LM it | PCG it | GPU s | GPU it/s | CPU it/s | CPU kind | max d_xy |",
          "|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        md.append("| %s | %d | %d | %s | %d | %d | %.3f | %.1f | %.2f | %s | %s |" % (
            r["workload"], r["poses"], r["edges"], r["policy"], r["lm_iters"], r["pcg_iters"], r["gpu_s"], r["gpu_it_s"],
            r["cpu_it_s"], r["cpu_kind"], ("%.1e" % r["max_dxy"]) if r["max_dxy"] is not None else "-"))
    open(os.path.join(ROOT, "profiles", tag + "_workloads.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
