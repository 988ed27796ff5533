"""CPU, world_size 2 over gloo: the algebra of the pose-id-range sharding.

Each rank owns a row range of H = J'J, evaluates every edge touching its rows with the oracle (cut edges on
both owners, cost counted where Edge::a lives), assembles its rows WITHOUT communication, and runs
block-Jacobi PCG with the solver's exchange steps (all-gather of the search direction, all-reduce of the dot
products).  The result must equal the single-process solution of the same damped normal equations, and the
shard bookkeeping must agree with libpgo's pgo_shard_plan."""
import os
import sys

import numpy as np
import pytest

from conftest import DATA, ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    import torch.distributed as dist
    import oracle as O
    import toy_robust_backend_slam_amd as P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = O.add_random_C(O.read_g2o(os.path.join(DATA, "MIT.g2o")), 40, 3)
        N, E = g.n_poses, g.n_edges
        rpr = -(-N // world)
        lo, hi = min(rank * rpr, N), min((rank + 1) * rpr, N)
        oa, ob = (g.ia >= lo) & (g.ia < hi), (g.ib >= lo) & (g.ib < hi)
        loc = np.nonzero(oa | ob)[0]
        plo, phi_, nl, ncut = P.shard_plan(N, g.ia, g.ib, world, rank)
        assert (plo, phi_, nl, ncut) == (lo, hi, len(loc), int(np.sum(oa ^ ob)))

        sub = O.Graph(g.pose_id, g.poses, g.ia[loc], g.ib[loc], g.meas[loc], g.info[loc], g.kind[loc])
        _, r, J = O.evaluate(sub, method=1)
        # cost: each edge once, on the owner of Edge::a
        rho_half = np.array([0.5 * O.huber(float(np.dot(e, e)), 0.01)[0] for e in O.evaluate(sub, method=1, apply_loss=False)[1]])
        cost = torch.tensor([float(np.sum(rho_half[oa[loc]]))], dtype=torch.float64)
        dist.all_reduce(cost)
        full_cost = O.evaluate(g, method=1, want_r=False, want_J=False)[0]
        assert abs(cost.item() - full_cost) < 1e-12 * full_cost

        # local Jacobian (3 E_loc x 3N), constant pose 0 dropped; owned rows of H and g need nothing remote
        EL = len(loc)
        rows = np.repeat(np.arange(3 * EL).reshape(EL, 3), 6, axis=1).reshape(-1)
        cols = np.concatenate([3 * sub.ia[:, None] + np.arange(3), 3 * sub.ib[:, None] + np.arange(3)], axis=1)
        cols = np.tile(cols, (1, 3)).reshape(-1)
        Jl = sp.csr_matrix((J.reshape(-1), (rows, cols)), shape=(3 * EL, 3 * N)).tolil()
        Jl[:, 0:3] = 0.0
        Jl = Jl.tocsr()
        own = np.arange(3 * lo, 3 * hi)
        H_rows = (Jl.T @ Jl).tocsr()[own, :]          # rows of the GLOBAL H: exact, because every edge touching
        g_rows = (Jl.T @ r.reshape(-1))[own]          # an owned row was evaluated locally
        d2 = np.maximum(H_rows[np.arange(len(own)), own].A1 if hasattr(H_rows[np.arange(len(own)), own], "A1")
                        else np.asarray(H_rows[np.arange(len(own)), own]).reshape(-1), 1e-6) / 1e2
        if lo == 0:
            d2[0:3] = 1.0
        A_rows = (H_rows + sp.csr_matrix((d2, (np.arange(len(own)), own)), shape=H_rows.shape)).tocsr()
        # block-Jacobi
        Minv = np.zeros((hi - lo, 3, 3))
        for i in range(hi - lo):
            blk = A_rows[3 * i:3 * i + 3, 3 * (lo + i):3 * (lo + i) + 3].toarray()
            Minv[i] = np.linalg.inv(blk)

        def allreduce(*vals):
            t = torch.tensor(vals, dtype=torch.float64)
            dist.all_reduce(t)
            return t.tolist()

        def allgather(v_own):
            pad = np.zeros(3 * rpr)
            pad[:len(v_own)] = v_own
            out = [torch.zeros(3 * rpr, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(pad))
            return np.concatenate([o.numpy() for o in out])[:3 * N]

        y = np.zeros(len(own))
        res = g_rows.copy()
        z = np.einsum("ijk,ik->ij", Minv, res.reshape(-1, 3)).reshape(-1)
        p = z.copy()
        rz, bb = allreduce(float(res @ z), float(res @ res))
        iters = 0
        for iters in range(1, 5000):
            p_full = allgather(p)
            Ap = A_rows @ p_full
            (pAp,) = allreduce(float(p @ Ap))
            alpha = rz / pAp
            y += alpha * p
            res -= alpha * Ap
            z = np.einsum("ijk,ik->ij", Minv, res.reshape(-1, 3)).reshape(-1)
            rz_new, rr = allreduce(float(res @ z), float(res @ res))
            if rr <= 1e-24 * bb:
                break
            p = z + (rz_new / rz) * p
            rz = rz_new
        y_full = allgather(y)

        # single-process reference for the same system
        _, rg, Jg = O.evaluate(g, method=1)
        rows = np.repeat(np.arange(3 * E).reshape(E, 3), 6, axis=1).reshape(-1)
        cols = np.concatenate([3 * g.ia[:, None] + np.arange(3), 3 * g.ib[:, None] + np.arange(3)], axis=1)
        cols = np.tile(cols, (1, 3)).reshape(-1)
        Jg = sp.csr_matrix((Jg.reshape(-1), (rows, cols)), shape=(3 * E, 3 * N)).tolil()
        Jg[:, 0:3] = 0.0
        Jg = Jg.tocsr()
        H = (Jg.T @ Jg).tocsc()
        D = np.maximum(H.diagonal(), 1e-6) / 1e2
        D[0:3] = 1.0
        y_ref = spla.splu((H + sp.diags(D)).tocsc()).solve(Jg.T @ rg.reshape(-1))
        err = float(np.abs(y_full - y_ref).max() / np.abs(y_ref).max())
        q.put((rank, iters, err, None))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, -1, float("inf"), traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sharded_pcg_equals_global_solve_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, iters, err, tb in out:
        assert tb is None, tb
        assert 0 < iters < 5000 and err < 1e-8, (rank, iters, err)
