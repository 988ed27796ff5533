#!/usr/bin/env python3
"""bench.py -- Gauss-Newton (LM) iterations/s of the HIP pose-graph backend on the synthetic
1M-pose Manhattan graph (BASELINE.json: "GN iterations/sec + edges/sec (residual+Jac) on 1M-pose
graph, 1/2/4/8 GPU").

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one LM iteration of the reference's solve (DCS-ceres/main.cpp:163) on the whole graph:
LM diagonal + chain preconditioner set-up, PCG on the normal equations (inexact steps: residual-norm
rtol 0.1, <= 500 iterations), model decrease, candidate cost (fused edge kernel, cost-only), accept /
reject, and on acceptance re-linearisation (fused residual+Jacobian kernel + assembly).  The graph is
sharded by pose-id range over the ranks (strong scaling: the problem is fixed, 1M poses).

Timing: W untimed LM iterations, then exactly K timed ones between barriers; that pass is repeated
`--passes` times from the initial poses (same trajectory every time) and the MEDIAN pass is reported
(`passes_ms_per_step` lists all of them).

Prints ONE JSON line on rank 0:
  roofline      the dominant kernel (block-CSR SpMV): algorithmic bytes / HIP-event launch time
  parity        GPU vs the CPU port of the same algorithm on the first `--cpu-iters` LM iterations of THIS
                workload (costs, accept/reject history, PCG counts, final translations)
  workloads     GN it/s of the other workloads north_star names: INTEL + 50 outliers (exact mode) and the
                synthetic 10k / 100k graphs (N = 1 only)
  cpu_baseline  the CPU port (oracle/pgo_oracle.c, -O2 -fopenmp) on the GPU box's host cores: all-core and
                one-thread samples of the same 1M-pose trajectory (N = 1 only)
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def kernels_digest():
    """identifies the kernel sources a committed PMC traffic figure belongs to"""
    h = hashlib.sha256()
    for f in ("kernels.hip.h", "coarse.hip.h", "solver_handle.hip.h", "solver_launch.hip", "solver_pcg.hip", "solver_lm.hip", "solver_create.hip"):
        h.update(open(os.path.join(ROOT, "toy-robust-backend-slam_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo; falls back to the logical count"""
    try:
        pairs, pk, co = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pk = line.split(":")[1].strip()
            elif line.startswith("core id"):
                co = line.split(":")[1].strip()
            elif not line.strip():
                if pk is not None and co is not None:
                    pairs.add((pk, co))
                pk = co = None
        if pk is not None and co is not None:
            pairs.add((pk, co))
        return len(pairs) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_quota():
    """cgroup CPU quota of this process in cores (None = unlimited / unknown)"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        return None


def main():
    # thread placement of the CPU-baseline legs (read by the OpenMP runtime when it starts)
    os.environ.setdefault("OMP_PLACES", "cores")
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--passes", type=int, default=5, help="timed passes of W + K LM iterations from the initial poses; the median is reported")
    ap.add_argument("--poses", type=int, default=1_000_000)
    ap.add_argument("--pcg-rtol", type=float, default=0.1)
    ap.add_argument("--pcg-max-iters", type=int, default=500)
    ap.add_argument("--pcg-block-poses", type=int, default=0, help="poses per dense block-Jacobi block, 0 = auto (GPU and CPU baseline)")
    ap.add_argument("--pcg-check-every", type=int, default=10,
                    help="PCG iterations per enqueued slice (one hipGraph replay at 1 GPU); the host checks the convergence flag "
                         "after the number of slices the previous solve makes likely, then after every slice")
    ap.add_argument("--pcg-chain-len", type=int, default=-1,
                    help="chain (block-tridiagonal) preconditioner over segments of this many poses, 0 = off, -1 = auto (GPU and CPU baseline)")
    ap.add_argument("--halo-exchange", type=int, default=-1,
                    help="N > 1: 1 = point-to-point halo exchange of the search direction, 0 = all-gather, -1 (default) = the p2p "
                         "exchange if a check against the all-gather on the first LM iterations agrees, else the all-gather")
    ap.add_argument("--pose-ordering", type=int, default=-1, help="internal locality ordering of the poses: 0 off, 1 on, -1 = the library's rule")
    ap.add_argument("--halo-overlap", type=int, default=0, help="N > 1: 1 = exchange on a second stream behind the owned-column SpMV")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--cpu-iters", type=int, default=8,
                    help="LM iterations of the all-core CPU baseline sample and of the parity check, capped at warmup + steps "
                         "(0 = skip both); ~1.3 s each at 1M poses on 16 threads")
    ap.add_argument("--cpu-iters-1t", type=int, default=1, help="1 = also the one-thread CPU sample (a bounded part of LM iteration 1), 0 = skip")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = sweep 16 / 64 / every physical core and take the fastest")
    ap.add_argument("--workloads", type=int, default=1, help="1 = also time INTEL+50 / 10k / 100k (N = 1 only)")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--k3", choices=["auto", "pipelined"], default="auto",
                    help="A/B switch: 'pipelined' keeps the persistent software-pipelined product kernel on large graphs (test hook spmv_pipe = 2)")
    ap.add_argument("--layout", choices=["auto", "dense"], default="auto",
                    help="A/B switch: 'dense' keeps the dense incidence layout on large graphs (test hook pad_tiles = 0)")
    ap.add_argument("--pcg-loop", choices=["auto", "one-reduction"], default="auto",
                    help="A/B switch: 'one-reduction' runs the several-rank PCG loop (Chronopoulos-Gear) on one rank too (test hook single_reduction = 1)")
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
    if args.k3 == "pipelined":
        P.set_knob("spmv_pipe", 2)
    if args.layout == "dense":
        P.set_knob("pad_tiles", 0)
    if args.pcg_loop == "one-reduction":
        P.set_knob("single_reduction", 1)
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

    def options(**kw):
        base = dict(method=1, max_iters=W + K, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0, pcg_rtol=args.pcg_rtol,
                    pcg_max_iters=args.pcg_max_iters, pcg_block_poses=args.pcg_block_poses, pcg_chain_len=args.pcg_chain_len,
                    halo_exchange=max(0, args.halo_exchange), halo_overlap=args.halo_overlap, pose_ordering=args.pose_ordering,
                    pcg_check_every=min(max(1, args.pcg_check_every), max(1, args.pcg_max_iters)), verbose=args.verbose if rank == 0 else 0)
        base.update(kw)
        return P.Options(**base)

    # Several ranks: the point-to-point halo exchange moves far fewer bytes than the all-gather, but the library keeps the
    # all-gather as its default until the p2p path has been checked against it on real peers -- so that check runs HERE,
    # on the first LM iterations of this workload, and the p2p path is timed only if every rank saw the same result.
    halo_check = None
    halo = max(0, args.halo_exchange)
    if world > 1 and args.halo_exchange < 0:
        res = []
        for hx in (0, 1):
            try:
                sc = P.Solver(g, options(max_iters=2, halo_exchange=hx, halo_overlap=0, verbose=0), comm, device=local_rank)
                smc = sc.solve()
                res.append((smc.final_cost, smc.total_pcg_iters, [r["step_ok"] for r in sc.iter_records()]))
                sc.close()
            except P.PgoError as e:
                res.append(("error", repr(e)))
        same = (len(res) == 2 and res[0][0] != "error" and res[1][0] != "error" and res[0][1:] == res[1][1:]
                and abs(res[0][0] - res[1][0]) <= 1e-12 * abs(res[0][0]))
        flag = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        halo = 1 if float(flag.cpu()[0]) > 0.5 else 0
        halo_check = {"all_gather": res[0], "p2p": res[1], "agree_on_every_rank": bool(halo), "timed": "p2p halo exchange" if halo else "all-gather"}
    opt = options(halo_exchange=halo)
    t_create = time.time()
    s = P.Solver(g, opt, comm, device=local_rank)
    t_create = time.time() - t_create
    info = s.info()  # what the library resolved the auto options to (no copy of its rules here)
    chain, blockp = info.pcg_chain_len, info.pcg_block_poses
    precond = ("block-tridiagonal %d-pose chain segments" % chain if chain else "dense %d-pose blocks" % blockp)

    x0 = np.array(g.poses)

    def timed_pass():
        """W untimed + exactly K timed LM iterations from the initial poses.  The minimiser can stop before max_iters
        = W + K only through MIN_RADIUS or FAILURE (a non-finite Jacobian at an accepted point: the reference's asin'
        singularity at |sin delta| = 1); it is then restarted from the initial poses INSIDE the timed region so that K
        iterations are always what is timed."""
        s.set_poses(x0)
        s.lm_begin()
        if W > 0:
            s.lm_step(W)
        recs = s.iter_records()
        n_prev = sum(1 for r in recs if r["iter"] > 0)
        barrier()
        t0 = time.perf_counter()
        timed, restarts, summ = [], 0, None
        while len(timed) < K:
            _, summ = s.lm_step(K - len(timed))
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
        return dt, timed, recs, summ, restarts

    passes = [timed_pass() for _ in range(max(1, args.passes))]
    order = sorted(range(len(passes)), key=lambda i: passes[i][0])
    dt, timed, recs, summ, restarts = passes[order[len(order) // 2]]   # the median pass

    # kernel-level numbers, measured live with HIP events on the solver's stream
    reps = args.kernel_reps
    k1 = s.bench_eval(reps, True)
    k1c = s.bench_eval(reps, False)
    k2 = s.bench_assemble(reps)
    k3 = s.bench_spmv(reps)
    pc = s.bench_precond(reps)
    all_dt = [p[0] for p in passes]
    vals = torch.tensor([dt, k1.ms_avg, k1c.ms_avg, k2.ms_avg, k3.ms_avg, pc.ms_avg] + all_dt, dtype=torch.float64,
                        device="cpu" if rehearsal else "cuda")
    if dist is not None:
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
    vals = [float(v) for v in vals.cpu()]
    dt_max, k1_ms, k1c_ms, k2_ms, k3_ms, pc_ms = vals[:6]
    all_dt = vals[6:]
    if dist is not None:  # the median of the per-pass maxima over the ranks
        dt_max = statistics.median_low(all_dt)

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
                            "outlier loops), DCS on, Huber 0.01, LM (Ceres policy) + block-Jacobi (%s) PCG, residual-norm rtol %.g <= %d it"
                            % (g.n_poses, n_edges, g.n_edges_of_kind(0), g.n_edges_of_kind(1), g.n_edges_of_kind(2),
                               precond, args.pcg_rtol, args.pcg_max_iters),
                "baseline_config": "configs[4] (synthetic 1M poses / ~4M edges, 10% outliers, sharded PCG)",
                "parallelism": "pose-id range shards x%d%s" % (world, "" if world == 1 else (
                    (", halo exchange" + (" overlapped" if info.halo_overlap else "")) if info.halo_exchange else ", all-gather")),
                "seed": 20260410,
            },
            "passes_ms_per_step": [1e3 * d / K for d in all_dt],
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
                "preconditioner apply (PCG start-up kernel, back to back)": {"ms": pc_ms, "GB/s": gbs(pc, pc_ms), "frac": gbs(pc, pc_ms) / HBM_PEAK_GBS},
            },
            "seconds": {"generate": t_gen, "create": t_create, "eval": summ.seconds_eval,
                        "assemble": summ.seconds_assemble, "linear": summ.seconds_linear,
                        "candidate": summ.seconds_candidate},
            "halo_exchange_check": halo_check,
            "handle": {k: v for k, v in s.info().as_dict().items() if k in ("pcg_block_poses", "pcg_chain_len", "chain_kernel", "n_tiles", "pose_ordering",
                                                                           "n_incidences", "halo_send_rows", "halo_recv_rows", "device_bytes",
                                                                           "host_enqueue_us_per_pcg_iter", "pcg_graph_replay", "pcg_single_reduction", "pcg_coarse_poses",
                                                                           "linear_solver")},
        }
        # HBM bytes per launch of k_spmv from the committed rocprofv3 --pmc passes -- only if they were taken on THESE
        # kernel sources (the file records a digest of them); otherwise null rather than a stale figure
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and args.poses == 1000000 and os.path.exists(pmc):
            try:
                tr = json.load(open(pmc))
                if tr.get("kernels_digest") == kernels_digest():
                    out["roofline"]["traffic"] = tr.get("k_spmv_bytes_per_launch")
                    out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build, kernels digest %s)" % tr["kernels_digest"]
                else:
                    out["roofline"]["traffic_source"] = "null: profiles/pmc_traffic.json was taken on other kernel sources (digest %s, this build %s)" % (
                        tr.get("kernels_digest"), kernels_digest())
            except Exception:
                pass

        # ---- the CPU port on the same trajectory: parity of the bench workload + the CPU baseline
        cpu_iters = min(args.cpu_iters, W + K)
        if world == 1 and cpu_iters > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            og = O.Graph(np.array(g.pose_ids), np.array(g.poses), np.array(g.ia), np.array(g.ib), np.array(g.meas),
                         np.array(g.info), np.array(g.kind))

            def port(iters, thr, variant=None, pcg_cap=None):
                oo = O.Options(method=1, max_iters=iters, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0,
                               pcg_rtol=args.pcg_rtol, pcg_max_iters=pcg_cap or args.pcg_max_iters, threads=thr,
                               pcg_block_poses=blockp, pcg_chain_len=chain)
                tc = time.perf_counter()
                r = O.lm_pcg(og, oo, variant=variant)
                return r, time.perf_counter() - tc

            def it_seconds(res):
                return sum(r["seconds"] for r in res.records if r["iter"] >= 1)

            # SURVEY 8(d) / BASELINE.md section 4: the port over the host's cores -- thread counts 16 / 64 / every physical
            # core (OMP_PLACES=cores, OMP_PROC_BIND=spread, set at the top of main) at the Makefile's -O2, then -O3
            # -march=native at the best count; one LM iteration each (the sample is bounded: the whole CPU leg ~30 s).
            # cpu_baseline.value = the fastest configuration, re-run over the parity iterations.
            phys = physical_cores()
            counts = [args.cpu_threads] if args.cpu_threads else sorted({c for c in (16, 64, phys) if c <= (os.cpu_count() or 1)} or {1})
            sweep = []
            for thr in counts:
                r1, _ = port(1, thr)
                sweep.append({"threads": thr, "flags": "-O2", "iter_per_s": 1.0 / it_seconds(r1)})
            best = max(sweep, key=lambda e: e["iter_per_s"])
            try:
                r1, _ = port(1, best["threads"], variant="O3-native")
                sweep.append({"threads": best["threads"], "flags": "-O3 -march=native", "iter_per_s": 1.0 / it_seconds(r1)})
            except Exception as e:
                sweep.append({"flags": "-O3 -march=native", "error": repr(e)})
            best = max((e for e in sweep if "iter_per_s" in e), key=lambda e: e["iter_per_s"])
            threads, variant = best["threads"], ("O3-native" if "O3" in best["flags"] else None)

            ores, tc = port(cpu_iters, threads, variant)
            it_s = it_seconds(ores)
            gpu_same = sum(r["seconds"] for r in recs if 1 <= r["iter"] <= cpu_iters)
            # GPU state after exactly cpu_iters iterations of the same trajectory
            s.set_poses(x0)
            s.lm_begin()
            s.lm_step(cpu_iters)
            grecs = [r for r in s.iter_records() if r["iter"] <= cpu_iters]
            gx = s.poses()
            orecs = [r for r in ores.records if r["iter"] <= cpu_iters]
            n_cmp = min(len(grecs), len(orecs))
            out["parity"] = {
                "against": "oracle/pgo_oracle.c pgo_oracle_lm_pcg (C port of the same LM + chain-preconditioned PCG; the oracle is "
                           "unpinned by the reference, which cannot be built here: no Ceres)",
                "iterations_compared": n_cmp - 1,
                "max_rel_cost": max(abs(a["cost"] - b["cost"]) / max(abs(b["cost"]), 1e-300) for a, b in zip(grecs[:n_cmp], orecs[:n_cmp])),
                "history_equal": [a["step_ok"] for a in grecs[:n_cmp]] == [b["step_ok"] for b in orecs[:n_cmp]],
                "max_pcg_iters_diff": max(abs(a["pcg_iters"] - b["pcg_iters"]) for a, b in zip(grecs[:n_cmp], orecs[:n_cmp])),
                "max_dxy": float(np.abs(gx[:, :2] - ores.poses[:, :2]).max()),
                "max_dtheta": float(np.abs(gx[:, 2] - ores.poses[:, 2]).max()),
            }
            out["cpu_baseline"] = {
                "value": cpu_iters / it_s,
                "unit": "iter/s",
                "cores": threads,
                "kind": "port",
                "flags": "gcc %s -fopenmp, OMP_PLACES=cores OMP_PROC_BIND=spread" % best["flags"],
                "cpu_model": cpu_model(),
                "host_cores_visible": os.cpu_count(),
                "host_physical_cores": phys,
                "host_cpu_quota": cpu_quota(),
                "sweep": sweep,
                "sample": "LM iterations 1..%d of the same 1M-pose workload (same options), oracle/pgo_oracle.c "
                          "pgo_oracle_lm_pcg with OpenMP, the fastest configuration of the sweep (one LM iteration per "
                          "configuration); %.1f s incl. first linearisation; PCG iterations %d; the GPU "
                          "took %.3f s for the same iterations" % (cpu_iters, tc, ores.total_pcg_iters, gpu_same),
                "direct_solve": "not timed here: a sparse direct factorisation (the reference's SPARSE_NORMAL_CHOLESKY) of these "
                                "graphs is dominated by the 10 % uniformly random loops -- measured once with the oracle's direct-solve "
                                "LM (scipy SuperLU, 1 thread) on the 10k-pose graph: 209 s per LM iteration (DESIGN.md section 6); the "
                                "port's PCG is the only CPU path that finishes at 100k / 1M",
            }
            if args.cpu_iters_1t > 0:
                # one thread (the reference runs Ceres with its default num_threads = 1): a bounded sample -- LM iteration 1 with
                # its PCG solve cut off after `cap` iterations; seconds per PCG iteration from that, the rest of the iteration
                # (linear-solve set-up, model, candidate) as measured, the PCG count of iteration 1 from the all-core run
                cap = 8
                o1, t1 = port(1, 1, None, pcg_cap=cap)
                rec1 = [r for r in o1.records if r["iter"] == 1][0]
                n_full = [r for r in ores.records if r["iter"] == 1][0]["pcg_iters"]
                oh, _ = port(1, 1, None, pcg_cap=1)
                rech = [r for r in oh.records if r["iter"] == 1][0]
                per_pcg = max(rec1["seconds"] - rech["seconds"], 0.0) / max(rec1["pcg_iters"] - rech["pcg_iters"], 1)
                est = rech["seconds"] + per_pcg * (n_full - rech["pcg_iters"])
                out["cpu_baseline"]["one_thread"] = {
                    "value": 1.0 / est, "unit": "iter/s", "cores": 1, "flags": "gcc -O2 -fopenmp",
                    "seconds_per_pcg_iteration": per_pcg,
                    "sample": "LM iteration 1 of the same workload with threads = 1, its PCG solve cut off after 1 and after %d "
                              "iterations (%.1f s + %.1f s incl. first linearisation): seconds per PCG iteration from the "
                              "difference, extrapolated to the %d PCG iterations LM iteration 1 needs" % (cap, _, t1, n_full)}
            # the reference's own build flavour (CMakeLists.txt:5-6: -fpermissive only, no -O level): the functor evaluation of
            # every edge (residual + Jacobian through Jets) at -O0 against -O2, one thread
            try:
                ev = {}
                for var in (None, "O0"):
                    L = O.lib(var)
                    x = np.ascontiguousarray(og.poses, np.float64)
                    rr, JJ = np.empty((og.n_edges, 3)), np.empty((og.n_edges, 18))
                    ia32, ib32 = np.ascontiguousarray(og.ia, np.int32), np.ascontiguousarray(og.ib, np.int32)
                    ms, kd = np.ascontiguousarray(og.meas, np.float64), np.ascontiguousarray(og.kind, np.uint8)
                    n_s = min(og.n_edges, 500000)
                    tc = time.perf_counter()
                    L.pgo_oracle_eval_w(og.n_poses, O._dp(x), n_s, O._ip(ia32), O._ip(ib32), O._dp(ms), None, O._bp(kd), 1, 0.5, 0.01, 1,
                                        O._dp(rr), O._dp(JJ), 1)
                    ev["-O2" if var is None else "-O0"] = (time.perf_counter() - tc) / n_s * 1e9
                out["cpu_baseline"]["functor_evaluation_ns_per_edge"] = dict(
                    ev, note="residual + 3x6 Jacobian (Jets through the reference's matrix expression) of %d edges, one thread; "
                             "-O0 is the reference's do_build.sh flavour (CMakeLists.txt:5-6 sets -fpermissive only)" % n_s)
            except Exception as e:
                out["cpu_baseline"]["functor_evaluation_ns_per_edge"] = {"error": repr(e)}

            # the reference's own workload (BASELINE configs[0]: INTEL + 50 outlier loops, Ceres SPARSE_NORMAL_CHOLESKY on
            # one thread): here a sparse direct factorisation IS the right CPU baseline -- the oracle's direct-solve LM
            # (scipy SuperLU standing in for Ceres' CHOLMOD path, 1 thread), whole 50-iteration solves
            if args.workloads:
                try:
                    gi_o = O.add_random_C(O.read_g2o(os.path.join(ROOT, "tests", "golden", "data", "INTEL.g2o")), 50, 1)
                    intel = {}
                    for m in (1, 0):
                        tc = time.perf_counter()
                        r_o = O.lm_direct(gi_o, O.Options(method=m))
                        intel["METHOD %d" % m] = {"gn_it_per_s": r_o.iterations / (time.perf_counter() - tc), "iterations": r_o.iterations,
                                                 "final_cost": r_o.final_cost}
                    out["cpu_baseline"]["intel_plus_50_direct_solve"] = dict(
                        intel, cores=1, kind="port", solver="oracle.lm_direct: the same LM policy, normal equations by scipy SuperLU")
                except Exception as e:
                    out["cpu_baseline"]["intel_plus_50_direct_solve"] = {"error": repr(e)}

        # ---- the other workloads north_star names (N = 1): whole 50-iteration solves, GN it/s = iterations / solve seconds
        if world == 1 and args.workloads:
            wl = {}
            data = os.path.join(ROOT, "tests", "golden", "data")
            golden = os.path.join(ROOT, "tests", "golden")

            def run(graph, **kw):
                sv = P.Solver(graph, P.Options(**kw), device=local_rank)
                x_init = np.array(graph.poses)
                sv.solve()                      # warm-up solve (graph capture, first touch)
                best = None
                for _ in range(3):
                    sv.set_poses(x_init)
                    sm = sv.solve()
                    if best is None or sm.seconds_total < best[0].seconds_total:
                        best = (sm, sv.poses())
                run.info = sv.info()
                sv.close()
                return best

            try:
                gi = P.ReadG2O(os.path.join(data, "INTEL.g2o"))
                gi.add_random_C(50, 1)
                # the library's default on this graph = the direct solve (odometry chain + low-rank Woodbury + refinement,
                # csrc/direct.hip.h), standing in for the reference's SPARSE_NORMAL_CHOLESKY; PCG to 1e-10 next to it
                for m in (1, 0):
                    for ls, label in ((0, "exact: direct chain + low-rank solve"), (1, "exact: PCG rtol 1e-10")):
                        sm, px = run(gi, method=m, linear_solver=ls)
                        ref = np.load(os.path.join(golden, "lm_INTEL_out50_m%d_poses.npy" % m))
                        wl["INTEL+50 METHOD %d (%s)" % (m, label)] = {
                            "gn_it_per_s": sm.iterations / sm.seconds_total, "iterations": sm.iterations, "pcg_iters": sm.total_pcg_iters,
                            "final_cost": sm.final_cost, "max_dxy_vs_oracle_direct_solve": float(np.abs(px[:, :2] - ref[:, :2]).max())}
                # BASELINE configs[2]: M3500 / MIT, DCS on / off (library defaults: MIT takes the direct solve, M3500 -- 1954 edges
                # outside its odometry chain -- PCG with dense 32-pose blocks)
                for name in ("MIT", "M3500", "FRH"):
                    gd = P.ReadG2O(os.path.join(data, name + ".g2o"))
                    for m in ((1, 0) if name != "FRH" else (1,)):
                        sm, px = run(gd, method=m, pcg_max_iters=400000)
                        solver = ("direct" if run.info.direct_switched_at == 0 else "PCG rtol 1e-10 and direct in turns, first change after LM iteration %d"
                                  % run.info.direct_switched_at) if (run.info.linear_solver == 2 or run.info.direct_switched_at) else "PCG rtol 1e-10"
                        if run.info.pcg_coarse_poses:
                            solver += ", two preconditioner levels (rigid-body modes of %d-pose aggregates, coarse order %d)" % (
                                run.info.pcg_coarse_poses, run.info.pcg_coarse_rank)
                        ref = np.load(os.path.join(golden, "lm_%s_out0_m%d_poses.npy" % (name, m)))
                        wl["%s METHOD %d (exact: %s)" % (name, m, solver)] = {
                            "gn_it_per_s": sm.iterations / sm.seconds_total, "iterations": sm.iterations, "pcg_iters": sm.total_pcg_iters,
                            "final_cost": sm.final_cost, "max_dxy_vs_oracle_direct_solve": float(np.abs(px[:, :2] - ref[:, :2]).max())}
            except Exception as e:  # the datasets are test fixtures: report, do not fail the bench line
                wl["datasets"] = {"error": repr(e)}
            def run_for(graph, seconds, max_lm=10 ** 9, **kw):
                """LM iterations from the initial poses for about `seconds` of wall clock or `max_lm` iterations, whichever comes
                first (one untimed iteration before: graph capture, first touch): [(elapsed s, cost, PCG iterations so far)]"""
                base = dict(method=1, max_iters=100000, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0)
                base.update(kw)
                sv = P.Solver(graph, P.Options(**base), device=local_rank)
                x_init = np.array(graph.poses)
                sv.lm_begin()
                sv.lm_step(1)
                sv.set_poses(x_init)
                sv.lm_begin()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                trace, pcg, done = [(0.0, sv.iter_records()[0]["cost"], 0)], 0, False
                while not done and time.perf_counter() - t0 < seconds and len(trace) <= max_lm:
                    done, _ = sv.lm_step(1)
                    r = sv.iter_records()[-1]
                    pcg += r["pcg_iters"]
                    trace.append((time.perf_counter() - t0, r["cost"] if r["step_ok"] == 1 else trace[-1][1], pcg))
                run_for.info = sv.info()
                sv.close()
                return trace

            graphs = {}
            for n in (10000, 100000):
                gs_ = graphs[n] = P.synth_manhattan(n, 4.0, 0.10, 20260410)
                sm, _ = run(gs_, method=1, max_iters=50, ftol=0.0, gtol=0.0, ptol=0.0, min_radius=0.0, pcg_rtol=args.pcg_rtol,
                            pcg_max_iters=args.pcg_max_iters, pcg_check_every=10)
                wl["synthetic %dk poses (inexact: rtol %.g)" % (n // 1000, args.pcg_rtol)] = {
                    "gn_it_per_s": sm.iterations / sm.seconds_total, "iterations": sm.iterations, "pcg_iters": sm.total_pcg_iters,
                    "n_edges": gs_.n_edges, "final_cost": sm.final_cost}
                # the reference's LM iteration is an EXACT solve (main.cpp:156, SPARSE_NORMAL_CHOLESKY): the same graph with the
                # linear systems solved to 1e-10 (library defaults otherwise: two preconditioner levels), the first 50 / 10 LM
                # iterations (a fixed count: later iterations need more PCG iterations, so a time budget would not compare)
                n_lm = 50 if n <= 10000 else 10
                tr = run_for(gs_, 8.0, max_lm=n_lm, pcg_rtol=1e-10, pcg_max_iters=1000000, pcg_check_every=50)
                info2 = run_for.info
                tr1 = run_for(gs_, 8.0, max_lm=n_lm, pcg_rtol=1e-10, pcg_max_iters=1000000, pcg_check_every=50, pcg_coarse_poses=0)
                wl["synthetic %dk poses (exact: rtol 1e-10)" % (n // 1000)] = {
                    "gn_it_per_s": (len(tr) - 1) / tr[-1][0], "iterations": len(tr) - 1, "pcg_iters": tr[-1][2], "seconds": tr[-1][0],
                    "final_cost": tr[-1][1], "coarse_aggregate_poses": info2.pcg_coarse_poses, "coarse_order": info2.pcg_coarse_rank,
                    "one_preconditioner_level": {"gn_it_per_s": (len(tr1) - 1) / tr1[-1][0], "iterations": len(tr1) - 1,
                                                 "pcg_iters": tr1[-1][2], "seconds": tr1[-1][0], "final_cost": tr1[-1][1]}}
            # What an iteration buys: the cost reached within fixed wall-clock budgets, for three forcing terms (residual-norm
            # tolerance of the PCG solve) -- an inexact iteration is cheap but moves less, and the headline's rtol must be an
            # EFFICIENT choice, not merely the one that maximises the iteration count.
            budgets = (0.5, 1.0, 2.0)
            ttc = {}
            for label, graph in (("100k", graphs[100000]), ("1M", g)):
                per = {}
                for rtol in (0.1, 0.01, 0.001):
                    tr = run_for(graph, budgets[-1], pcg_rtol=rtol, pcg_max_iters=20000, pcg_check_every=10 if rtol >= 0.1 else 50)
                    row = {"initial_cost": tr[0][1]}
                    for bd in budgets:
                        inside = [t for t in tr if t[0] <= bd]
                        row["%.1f s" % bd] = {"cost": inside[-1][1], "lm_iters": len(inside) - 1, "pcg_iters": inside[-1][2]}
                    per["rtol %g" % rtol] = row
                ttc[label] = per
            wl["time_to_cost"] = ttc
            out["workloads"] = wl
        print(json.dumps(out))
    s.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
