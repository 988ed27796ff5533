"""CPU restatement (numpy) of the direct linear solve of csrc/direct.hip.h -- odometry chain T factorised by the block
recurrence the kernel runs, every other edge as a low-rank term V'V through the Woodbury identity, one step of iterative
refinement -- against the oracle's sparse direct solve (SuperLU) of the same LM system.  Pins the algebra (splitting,
recurrence, segment-parallel sweeps with prefix products) independently of the GPU; the GPU path itself is checked in
tests/test_gpu_parity.py::test_direct_solve_*."""
import os

import numpy as np
import pytest
import scipy.linalg as sl
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import DATA


def _system(oracle, name, n_out, method, radius):
    """scaled Jacobian rows (free columns), LM diagonal, right-hand side -- as oracle.lm_direct builds them at iteration 1"""
    g = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
    if n_out:
        g = oracle.add_random_C(g, n_out, 1)
    N, E = g.n_poses, g.n_edges
    cost, r, J = oracle.evaluate(g, g.poses, method, 0.5, 0.01, True, True, True, 1, False)
    rows = np.repeat(np.arange(3 * E).reshape(E, 3), 6, axis=1).reshape(-1)
    cols = np.concatenate([3 * g.ia[:, None] + np.arange(3), 3 * g.ib[:, None] + np.arange(3)], axis=1)
    cols = np.tile(cols, (1, 3)).reshape(-1).astype(np.int64)
    A = sp.csr_matrix((J.reshape(-1), (rows, cols)), shape=(3 * E, 3 * N))[:, 3:].tocsc()   # pose 0 constant
    s = 1.0 / (1.0 + np.sqrt(np.asarray(A.multiply(A).sum(axis=0)).reshape(-1)))
    As = (A @ sp.diags(s)).tocsr()
    H = (As.T @ As).tocsc()
    D2 = np.clip(H.diagonal(), 1e-6, 1e32) / radius
    gs = s * (A.T @ r.reshape(-1))
    return g, As, H, D2, gs


def _chain_split(g):
    chain = -np.ones(g.n_poses, np.int64)
    for e, (a, b) in enumerate(zip(g.ia, g.ib)):
        lo, hi = min(a, b), max(a, b)
        if hi == lo + 1 and chain[lo] < 0:
            chain[lo] = e
    is_chain = np.zeros(g.n_edges, bool)
    is_chain[chain[chain >= 0]] = True
    return is_chain


def _factor(T):
    """k_dlr_factor: W_i = C_i S_{i-1}^-1, S_i = M_i - W_i C_i'"""
    n = T.shape[0] // 3
    Td = T.toarray()
    W, Sinv = [np.zeros((3, 3))] * n, [None] * n
    Sinv[0] = np.linalg.inv(Td[:3, :3])
    for i in range(1, n):
        C = Td[3 * i:3 * i + 3, 3 * i - 3:3 * i]
        W[i] = C @ Sinv[i - 1]
        Sinv[i] = np.linalg.inv(Td[3 * i:3 * i + 3, 3 * i:3 * i + 3] - W[i] @ C.T)
    return W, Sinv


def _solve_segmented(W, Sinv, B, nseg=32):
    """k_dlr_prefix / _fwd / _mid / _fix: sweeps cut into segments, joined through the prefix products G, Gb"""
    n = len(W)
    L = -(-n // nseg)
    segs = [(s0, min(n, s0 + L)) for s0 in range(0, n, L)]
    Wn = W + [np.zeros((3, 3))]
    G, Gb = [None] * n, [None] * n
    for i0, i1 in segs:
        g = np.eye(3)
        for i in range(i0, i1):
            g = -W[i] @ g
            G[i] = g
        g = np.eye(3)
        for i in range(i1 - 1, i0 - 1, -1):
            g = -Wn[i + 1].T @ g
            Gb[i] = g
    X = B.copy().reshape(n, 3, -1)
    E = []
    for i0, i1 in segs:                       # local forward sweeps
        t = np.zeros_like(X[0])
        for i in range(i0, i1):
            t = X[i] - W[i] @ t
            X[i] = t
        E.append(t)
    tin, E2 = np.zeros_like(X[0]), []
    tins = []
    for q, (i0, i1) in enumerate(segs):
        tins.append(tin)
        tin = E[q] + G[i1 - 1] @ tin
    for q, (i0, i1) in enumerate(segs):       # true t on the fly, local backward sweeps
        z = np.zeros_like(X[0])
        for i in range(i1 - 1, i0 - 1, -1):
            t = X[i] + G[i] @ tins[q]
            z = Sinv[i] @ t - Wn[i + 1].T @ z
            X[i] = z
        E2.append(z)
    xin = np.zeros_like(X[0])
    for q in range(len(segs) - 1, -1, -1):    # incoming x from the right
        i0, i1 = segs[q]
        if q < len(segs) - 1:
            for i in range(i0, i1):
                X[i] = X[i] + Gb[i] @ xin
        xin = E2[q] + Gb[i0] @ xin
    return X.reshape(3 * n, -1)


@pytest.mark.parametrize("radius", [1e4, 1e12])
@pytest.mark.parametrize("method", [0, 1])
def test_woodbury_chain_plus_low_rank_equals_sparse_direct_solve(oracle, method, radius):
    g, As, H, D2, gs = _system(oracle, "INTEL", 50, method, radius)
    y_ref = spla.splu((H + sp.diags(D2)).tocsc()).solve(gs)
    is_chain = _chain_split(g)
    rows_c = np.repeat(is_chain, 3)
    Ac, V = As[rows_c], As[~rows_c].toarray()
    T = (Ac.T @ Ac + sp.diags(D2)).tocsc()
    assert abs(T - (H + sp.diags(D2) - sp.csr_matrix(V.T @ V))).max() < 1e-12      # H + D'D = T + V'V
    K = V.shape[0]
    assert K == 3 * (256 + 50)
    W, Sinv = _factor(T)
    ZT = _solve_segmented(W, Sinv, np.concatenate([V.T, gs[:, None]], axis=1))
    assert np.abs(T @ ZT - np.concatenate([V.T, gs[:, None]], axis=1)).max() < 1e-7 * max(1.0, np.abs(ZT).max())
    Z, t = ZT[:, :K], ZT[:, K]
    cf = sl.cho_factor(np.eye(K) + V @ Z)
    y = t - Z @ sl.cho_solve(cf, V @ t)
    e0 = np.linalg.norm(y - y_ref) / np.linalg.norm(y_ref)
    res = gs - (H @ y + D2 * y)                                                     # one refinement step (GPU: k_spmv)
    tt = _solve_segmented(W, Sinv, res[:, None])[:, 0]
    y = y + tt - Z @ sl.cho_solve(cf, V @ tt)
    e1 = np.linalg.norm(y - y_ref) / np.linalg.norm(y_ref)
    print(f"METHOD {method} radius {radius:.0e}: rank {K}, relative error {e0:.1e} -> {e1:.1e} after one refinement step")
    assert e0 < 1e-4 and e1 < 1e-8


@pytest.mark.parametrize("radius", [1e4, 1e16])
def test_separators_restore_the_whole_chain(oracle, radius):
    """k_dlr_sep_*: the chain is factorised in 4 pieces side by side (nested dissection with 3 separator poses).  With
    Tt = T without the separators' couplings (what the sweeps invert), B those couplings, Ms the separators' diagonal blocks:
    x_s = S^-1 (Ms z_s - B' z_p), x_p = z_p - Y x_s, z = Tt^-1 r, Y = Tt^-1 B, S = Ms - B' Y (SPD, order 9)"""
    g, As, H, D2, gs = _system(oracle, "INTEL", 50, 1, radius)
    rows_c = np.repeat(_chain_split(g), 3)
    Ac = As[rows_c]
    Td = (Ac.T @ Ac + sp.diags(D2)).toarray()
    n = Td.shape[0] // 3
    seps = [(k * n) // 4 for k in (1, 2, 3)]
    Tt = Td.copy()
    B = np.zeros((3 * n, 9))
    srow = np.concatenate([np.arange(3 * s, 3 * s + 3) for s in seps])
    for j, s in enumerate(seps):
        B[3 * s - 3:3 * s, 3 * j:3 * j + 3] = Td[3 * s - 3:3 * s, 3 * s:3 * s + 3]          # C_s'
        B[3 * s + 3:3 * s + 6, 3 * j:3 * j + 3] = Td[3 * s + 3:3 * s + 6, 3 * s:3 * s + 3]  # C_{s+1}
        for (r0, c0) in ((s, s - 1), (s - 1, s), (s + 1, s), (s, s + 1)):
            Tt[3 * r0:3 * r0 + 3, 3 * c0:3 * c0 + 3] = 0.0
    assert np.linalg.eigvalsh(Tt)[0] > 1e-9                     # the pieces stay anchored: every diagonal block keeps all its edges
    lu = spla.splu(sp.csc_matrix(Tt))
    rhs = np.random.default_rng(1).standard_normal((3 * n, 4))
    rhs[:, 0] = gs
    z, Y = lu.solve(rhs), lu.solve(B)
    assert np.abs(Y[srow]).max() == 0.0
    Ms = Td[np.ix_(srow, srow)]
    S = Ms - B.T @ Y
    assert np.linalg.eigvalsh(0.5 * (S + S.T))[0] > 0.0
    w = np.linalg.solve(S, Ms @ z[srow] - B.T @ z)
    x = z - Y @ w
    x[srow] = w
    ref = spla.splu(sp.csc_matrix(Td)).solve(rhs)
    err = np.abs(x - ref).max() / np.abs(ref).max()
    print(f"radius {radius:.0e}: 3 separators, Schur complement condition {np.linalg.cond(S):.1e}, relative error of T^-1 r {err:.1e}")
    assert err < 1e-7
