#!/usr/bin/env python3
"""bench.py -- Gauss-Newton (LM) iterations/s of the HIP pose-graph backend on the synthetic
1M-pose Manhattan graph (BASELINE.json: "GN iterations/sec + edges/sec (residual+Jac) on 1M-pose
graph, 1/2/4/8 GPU").

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one LM iteration of the reference's solve (DCS-ceres/main.cpp:163) on the whole graph:
LM diagonal + block-Jacobi preconditioner, PCG on the normal equations (Ceres' inexact-step defaults:
eta = 0.1, <= 500 iterations), model-decrease, candidate cost (fused edge kernel, cost-only), accept /
reject, and on acceptance re-linearisation (fused residual+Jacobian kernel + assembly).  The graph is
sharded by pose-id range over the ranks (strong scaling: the problem is fixed, 1M poses).

Prints ONE JSON line on rank 0.  The `roofline` object is for the dominant kernel (block-CSR SpMV);
`cpu_baseline` is the CPU oracle ("port": same algorithm, C + OpenMP) timed on rank 0 at N=1 on a
bounded sample (the first LM iteration(s) of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--poses", type=int, default=1_000_000)
    ap.add_argument("--pcg-rtol", type=float, default=0.1)
    ap.add_argument("--pcg-max-iters", type=int, default=500)
    ap.add_argument("--pcg-block-poses", type=int, default=0, help="poses per dense block-Jacobi block, 0 = auto (GPU and CPU baseline)")
    ap.add_argument("--pcg-check-every", type=int, default=100,
                    help="PCG iterations enqueued (as one hipGraph replay at 1 GPU) between two host checks of the convergence flag")
    ap.add_argument("--pcg-chain-len", type=int, default=-1,
                    help="chain (block-tridiagonal) preconditioner over segments of 64 poses: 64 = on, 0 = off, -1 = auto (GPU and CPU baseline)")
    ap.add_argument("--halo-exchange", type=int, default=1, help="N > 1: 1 = point-to-point halo exchange of the search direction, 0 = all-gather")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--cpu-iters", type=int, default=8,
                    help="LM iterations of the CPU baseline sample, capped at warmup + steps (0 = skip); ~1 s each at 1M poses")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, cores)")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--comm", choices=["rccl", "shm"], default="rccl",
                    help="shm = rehearsal on ONE GPU: gloo process group + host-staged shared-memory communicator, all ranks on cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    import toy_robust_backend_slam_amd as P

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP backend has no CPU path", file=sys.stderr)
        sys.exit(3)
    rehearsal = args.comm == "shm"
    if rehearsal:
        local_rank = 0  # every rank shares cuda:0 (RCCL would refuse duplicate devices)
    torch.cuda.set_device(local_rank)
    dist = None
    comm = None
    force_dist = os.environ.get("PGO_BENCH_FORCE_DIST") == "1"  # exercise the distributed path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            comm = P.Comm.shm("pgo_bench_%s" % os.environ["MASTER_PORT"], rank, world, 0)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(P.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            comm = P.Comm.rccl(bytes(uid.cpu().numpy().tobytes()), rank, world, local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_gen = time.time()
    g = P.synth_manhattan(args.poses, 4.0, 0.10, 20260410)
    t_gen = time.time() - t_gen
    K, W = args.steps, args.warmup
    opt = P.Options(method=1, max_iters=W + K, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0, pcg_rtol=args.pcg_rtol,
                    pcg_max_iters=args.pcg_max_iters, pcg_block_poses=args.pcg_block_poses, pcg_chain_len=args.pcg_chain_len,
                    halo_exchange=args.halo_exchange,
                    pcg_check_every=min(max(1, args.pcg_check_every), max(1, args.pcg_max_iters)), verbose=args.verbose if rank == 0 else 0)
    # the preconditioner the library resolves for these options (pgo_internal.h resolve_chain_len / resolve_block_poses)
    chain = args.pcg_chain_len if args.pcg_chain_len >= 0 else (64 if (args.pcg_block_poses <= 0 and g.n_poses > 50000) else 0)
    blockp = args.pcg_block_poses if args.pcg_block_poses > 0 else (32 if g.n_poses <= 8192 else 4)
    precond = ("block-tridiagonal 64-pose chain segments" if chain else "dense %d-pose blocks" % blockp)
    t_create = time.time()
    s = P.Solver(g, opt, comm, device=local_rank)
    t_create = time.time() - t_create

    x0 = np.array(g.poses)
    s.lm_begin()
    if W > 0:
        s.lm_step(W)
    recs = s.iter_records()
    n_prev = sum(1 for r in recs if r["iter"] > 0)
    barrier()
    t0 = time.perf_counter()
    # exactly K LM iterations.  The minimiser can stop before max_iters = W + K only through MIN_RADIUS or FAILURE (a
    # non-finite Jacobian at an accepted point: the reference's asin' singularity at |sin delta| = 1); it is then restarted
    # from the initial poses INSIDE the timed region so that K iterations are always what is timed.
    timed, restarts = [], 0
    while len(timed) < K:
        done, summ = s.lm_step(K - len(timed))
        now = s.iter_records()
        n_now = sum(1 for r in now if r["iter"] > 0)
        new = now[len(now) - (n_now - n_prev):] if n_now > n_prev else []
        if restarts == 0:
            recs = now
        timed += new
        n_prev = n_now
        if len(timed) < K:
            restarts += 1
            if restarts > 8 and not new:
                raise RuntimeError("bench: the solver makes no progress: %r" % (summ.as_dict(),))
            s.set_poses(x0)
            s.lm_begin()
            n_prev = 0
    barrier()
    dt = time.perf_counter() - t0
    assert len(timed) == K, (len(timed), K, summ.as_dict())

    # kernel-level numbers, measured live with HIP events on the solver's stream
    reps = args.kernel_reps
    k1 = s.bench_eval(reps, True)
    k1c = s.bench_eval(reps, False)
    k2 = s.bench_assemble(reps)
    k3 = s.bench_spmv(reps)
    vals = torch.tensor([dt, k1.ms_avg, k1c.ms_avg, k2.ms_avg, k3.ms_avg], dtype=torch.float64,
                        device="cpu" if rehearsal else "cuda")
    if dist is not None:
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
    dt_max, k1_ms, k1c_ms, k2_ms, k3_ms = [float(v) for v in vals.cpu()]

    if rank == 0:
        def gbs(stats, ms):
            return stats.algorithmic_bytes / (ms * 1e-3) / 1e9

        n_edges = g.n_edges
        out = {
            "metric": "gn_iterations_per_sec",
            "value": K / dt_max,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * dt_max / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic Manhattan world, %d poses / %d edges (%d odometry, %d closure, %d bogus = 10%% "
                            "outlier loops), DCS on, Huber 0.01, LM (Ceres policy) + block-Jacobi (%s) PCG rtol %.g <= %d it"
                            % (g.n_poses, n_edges, g.n_edges_of_kind(0), g.n_edges_of_kind(1), g.n_edges_of_kind(2),
                               precond, args.pcg_rtol, args.pcg_max_iters),
                "baseline_config": "configs[4] (synthetic 1M poses / ~4M edges, 10% outliers, sharded PCG)",
                "parallelism": "pose-id range shards x%d%s" % (world, "" if world == 1 else (", halo exchange" if args.halo_exchange else ", all-gather")),
                "seed": 20260410,
            },
            "edges_per_sec": n_edges / (k1_ms * 1e-3),
            "pcg_iters_per_step": sum(r["pcg_iters"] for r in timed) / K,
            "accepted_steps": sum(1 for r in timed if r["step_ok"] == 1),
            "cost_first_last": [recs[0]["cost"], summ.final_cost],
            "restarts": restarts,
            "roofline": {
                "kernel": "k_spmv (block-CSR 3x3 SpMV, fp64)",
                "bound": "hbm",
                "achieved": gbs(k3, k3_ms),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": gbs(k3, k3_ms) / HBM_PEAK_GBS,
                "traffic": None,
                "ms": k3_ms,
                "algorithmic_bytes": k3.algorithmic_bytes,
            },
            "kernels": {
                "k_edge_eval<jac>": {"ms": k1_ms, "GB/s": gbs(k1, k1_ms), "frac": gbs(k1, k1_ms) / HBM_PEAK_GBS,
                                     "edges_per_s": k1.units / (k1_ms * 1e-3)},
                "k_edge_eval<cost>": {"ms": k1c_ms, "GB/s": gbs(k1c, k1c_ms), "frac": gbs(k1c, k1c_ms) / HBM_PEAK_GBS},
                "k_assemble": {"ms": k2_ms, "GB/s": gbs(k2, k2_ms), "frac": gbs(k2, k2_ms) / HBM_PEAK_GBS},
                "k_spmv": {"ms": k3_ms, "GB/s": gbs(k3, k3_ms), "frac": gbs(k3, k3_ms) / HBM_PEAK_GBS},
            },
            "seconds": {"generate": t_gen, "create": t_create, "eval": summ.seconds_eval,
                        "assemble": summ.seconds_assemble, "linear": summ.seconds_linear,
                        "candidate": summ.seconds_candidate},
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and args.poses == 1000000 and os.path.exists(pmc):  # HBM bytes per launch of the whole-matrix k_spmv, from the committed rocprofv3 --pmc passes
            try:
                out["roofline"]["traffic"] = json.load(open(pmc)).get("k_spmv_bytes_per_launch")
            except Exception:
                pass
        cpu_iters = min(args.cpu_iters, W + K)   # the sample is the start of the same LM trajectory the GPU just ran
        if world == 1 and cpu_iters > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            og = O.Graph(np.array(g.pose_ids), np.array(g.poses), np.array(g.ia), np.array(g.ib), np.array(g.meas),
                         np.array(g.info), np.array(g.kind))
            oo = O.Options(method=1, max_iters=cpu_iters, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0,
                           pcg_rtol=args.pcg_rtol, pcg_max_iters=args.pcg_max_iters, threads=threads,
                           pcg_block_poses=blockp, pcg_chain_len=chain)
            tc = time.perf_counter()
            ores = O.lm_pcg(og, oo)
            tc = time.perf_counter() - tc
            it_s = sum(r["seconds"] for r in ores.records if r["iter"] >= 1)
            gpu_same = sum(r["seconds"] for r in recs if 1 <= r["iter"] <= cpu_iters)
            out["cpu_baseline"] = {
                "value": cpu_iters / it_s,
                "unit": "iter/s",
                "cores": threads,
                "kind": "port",
                "sample": "LM iterations 1..%d of the same 1M-pose workload (same options), oracle/pgo_oracle.c "
                          "pgo_oracle_lm_pcg with OpenMP; %.1f s incl. first linearisation; PCG iterations %d; the GPU "
                          "took %.3f s for the same iterations" % (cpu_iters, tc, ores.total_pcg_iters, gpu_same),
            }
        print(json.dumps(out))
    s.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
