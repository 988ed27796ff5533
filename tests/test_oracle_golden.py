"""CPU: the oracle (oracle/) against known-answer values.

PARITY UNPINNED with respect to the reference binary: the reference ships no tests and cannot be built
here.  What pins the oracle instead:
  * SURVEY.md section 8(c) known-answer values, derived there from the reference's formulas with an
    independent closed-form numpy restatement (the oracle uses forward-mode Jets through the
    reference's matrix expression, so the two derivations share no code);
  * finite differences of the oracle's own residual;
  * the committed fixtures under tests/golden/ (regression of the restatement).
"""
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN, DATASETS

# initial cost 1/2 sum rho at the file poses, METHOD 0 / METHOD 1, and #edges with psi < 1
KNOWN = {
    "INTEL": (3.082223216921e+01, 2.471759982854e+00, 251),
    "MIT": (1.459953365886e+01, 1.959203269806e-01, 20),
    "CSAIL": (4.063356725792e+00, 9.610536556244e-01, 97),
    "FR079": (4.870069294740e-01, 4.865471483015e-01, 1),
    "FRH": (2.254308375633e-03, 2.254308375633e-03, 0),
    "M3500": (4.989870673947e+01, 1.163382523718e+01, 992),
}


@pytest.mark.parametrize("name", DATASETS)
def test_initial_costs_known_answers(oracle, name):
    g = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
    c0 = oracle.evaluate(g, method=0, want_r=False, want_J=False)[0]
    c1 = oracle.evaluate(g, method=1, want_r=False, want_J=False)[0]
    e0, e1, npsi = KNOWN[name]
    assert c0 == pytest.approx(e0, rel=1e-11)
    assert c1 == pytest.approx(e1, rel=1e-11)
    _, r0, _ = oracle.evaluate(g, method=0, apply_loss=False, want_J=False)
    _, r1, _ = oracle.evaluate(g, method=1, apply_loss=False, want_J=False)
    assert int(np.sum(np.abs(r1 - r0).max(axis=1) > 0)) == npsi
    fx = json.load(open(os.path.join(GOLDEN, "initial_costs.json")))[name]
    assert fx["cost_method0"] == pytest.approx(c0, rel=1e-13) and fx["cost_method1"] == pytest.approx(c1, rel=1e-13)
    assert fx["n_poses"] == g.n_poses and fx["n_edges"] == g.n_edges


def test_intel_first_closure_edge(oracle):
    """file edge #1227 (19 -> 166)"""
    g = oracle.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    k = 1227
    assert (g.ia[k], g.ib[k]) == (19, 166) and g.kind[k] == 1
    np.testing.assert_allclose(g.meas[k], [-2.459689, 0.241111, 0.2528])
    e, J = oracle.edge(g.poses[19], g.poses[166], g.meas[k], False)
    np.testing.assert_allclose(e, [0.3442233556645796, -2.1712824454602253, 0.01997599999999999], rtol=1e-12)
    e, J = oracle.edge(g.poses[19], g.poses[166], g.meas[k], True)
    np.testing.assert_allclose(e, [0.14905834188090758, -0.9402260356522534, 0.00865016678390194], rtol=1e-12)
    np.testing.assert_allclose(J[0], [-0.42651265331100036, -0.032300575930653677, -0.44002998504004065,
                                      0.42651265331100036, 0.032300575930653677, 0.0], rtol=1e-11, atol=1e-15)
    e, J = oracle.edge(g.poses[3], g.poses[4], g.meas[3], False)
    np.testing.assert_allclose(J[:, 2], [-0.00553499534995826, -0.630064658973566, -1.0], rtol=1e-11)


def test_asin_fold(oracle):
    """theta error is asin(sin d): folds to [-pi/2, pi/2] with derivative sign(cos d)"""
    e, J = oracle.edge([0, 0, 0], [0, 0, 2.5], [0, 0, 0], False)
    assert e[2] == pytest.approx(np.pi - 2.5, rel=1e-13)
    assert J[2, 5] == pytest.approx(-1.0, rel=1e-12) and J[2, 2] == pytest.approx(1.0, rel=1e-12)
    e, J = oracle.edge([0, 0, 0], [0, 0, 1.0], [0, 0, 0], False)
    assert J[2, 5] == pytest.approx(1.0, rel=1e-12)


@pytest.mark.parametrize("dcs", [False, True])
def test_jet_jacobian_matches_finite_differences(oracle, dcs):
    rng = np.random.default_rng(5)
    for _ in range(50):
        P1, P2 = rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 3)
        m = rng.uniform(-2, 2, 3)
        e, J = oracle.edge(P1, P2, m, dcs)
        x = np.concatenate([P1, P2])
        num = np.zeros((3, 6))
        h = 1e-6
        for k in range(6):
            xp, xm = x.copy(), x.copy()
            xp[k] += h
            xm[k] -= h
            num[:, k] = (oracle.edge(xp[:3], xp[3:], m, dcs, jac=False) - oracle.edge(xm[:3], xm[3:], m, dcs, jac=False)) / (2 * h)
        ed = oracle.edge(P1, P2, m, dcs, jac=False)
        np.testing.assert_allclose(ed, e, rtol=1e-13, atol=1e-15)   # T=double and T=Jet instantiations agree
        if abs(abs(np.sin(P2[2] - P1[2] - m[2])) - 1) < 1e-3:
            continue
        np.testing.assert_allclose(J, num, rtol=2e-6, atol=2e-7)


def test_dcs_is_identity_inside_phi(oracle):
    """psi == 1 with zero derivative when ex^2 + ey^2 <= phi (Jet min picks T(1.0))"""
    P1, P2, m = [0.1, 0.2, 0.3], [0.5, 0.1, 0.4], [0.3, -0.1, 0.1]
    e0, J0 = oracle.edge(P1, P2, m, False)
    e1, J1 = oracle.edge(P1, P2, m, True)
    assert e0[0] ** 2 + e0[1] ** 2 < 0.5
    np.testing.assert_array_equal(e0, e1)
    np.testing.assert_array_equal(J0, J1)


def test_huber(oracle):
    a = 0.01
    np.testing.assert_allclose(oracle.huber(0.5e-4, a), [0.5e-4, 1.0, 0.0])
    s = 4.0
    rho = oracle.huber(s, a)
    np.testing.assert_allclose(rho, [2 * a * 2.0 - a * a, a / 2.0, -(a / 2.0) / (2 * s)])


def test_glibc_rand_restatement(oracle):
    r = oracle.GlibcRand(1)
    assert [r.rand() for _ in range(5)] == [1804289383, 846930886, 1681692777, 1714636915, 1957747793]


def test_injector_known_answers(oracle):
    g = oracle.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    g2 = oracle.add_random_C(g, 50, 1)
    assert list(zip(g2.ia[-50:][:4].tolist(), g2.ib[-50:][:4].tolist())) == [(35, 162), (1175, 1086), (422, 711), (46, 660)]
    assert np.all(g2.meas[-50:] == 0.0) and np.all(g2.kind[-50:] == 2)
    c = oracle.evaluate(g2, method=1, want_r=False, want_J=False)[0]
    assert c == pytest.approx(2.969102e+00, rel=1e-6)
    fx = json.load(open(os.path.join(GOLDEN, "intel_bogus.json")))
    for seed in (1, 2, 3):
        gg = oracle.add_random_C(g, 50, seed)
        assert gg.ia[-50:].tolist() == fx[str(seed)]["a"] and gg.ib[-50:].tolist() == fx[str(seed)]["b"]


def test_edge_fixture_regression(oracle):
    for rec in json.load(open(os.path.join(GOLDEN, "intel_edges.json"))):
        for dcs in (0, 1):
            e, J = oracle.edge(rec["P1"], rec["P2"], rec["meas"], bool(dcs))
            np.testing.assert_allclose(e, rec["e%d" % dcs], rtol=1e-13, atol=1e-16)
            np.testing.assert_allclose(J.reshape(-1), rec["J%d" % dcs], rtol=1e-13, atol=1e-16)


def test_lm_direct_intel_matches_fixture_and_baseline_md(oracle):
    """INTEL METHOD 1, 50 srand(1) outliers: 2.9691 -> 0.66796 at the 50-iteration cap"""
    g = oracle.add_random_C(oracle.read_g2o(os.path.join(DATA, "INTEL.g2o")), 50, 1)
    res = oracle.lm_direct(g, oracle.Options(method=1))
    assert res.termination == 4 and res.iterations == 50
    assert res.initial_cost == pytest.approx(2.9691, rel=1e-4) and res.final_cost == pytest.approx(0.66796, rel=1e-4)
    fx = json.load(open(os.path.join(GOLDEN, "lm_INTEL_out50_m1.json")))
    assert res.final_cost == pytest.approx(fx["final_cost"], rel=1e-9)
    ref = np.load(os.path.join(GOLDEN, "lm_INTEL_out50_m1_poses.npy"))
    assert np.abs(res.poses - ref).max() < 1e-7
    for a, b in zip(res.records, fx["records"]):
        assert a["step_ok"] == b["step_ok"] and a["radius"] == pytest.approx(b["radius"], rel=1e-9)


def test_lm_pcg_port_tracks_direct_solve(oracle):
    """the C "port" (block-Jacobi PCG) follows the direct-solve trajectory when solved tightly"""
    g = oracle.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    a = oracle.lm_direct(g, oracle.Options(method=1, max_iters=6))
    b = oracle.lm_pcg(g, oracle.Options(method=1, max_iters=6, pcg_rtol=1e-12, pcg_max_iters=100000, threads=4))
    assert a.iterations == b.iterations == 6
    assert np.abs(a.poses - b.poses).max() < 1e-8
    assert a.final_cost == pytest.approx(b.final_cost, rel=1e-9)


def test_lm_policy_rejected_step_shrinks_radius(oracle):
    """a huge initial radius on a hard start forces rejected steps: radius /= 2, 4, ... (LevenbergMarquardtStrategy)"""
    g = oracle.read_g2o(os.path.join(DATA, "M3500.g2o"))
    res = oracle.lm_direct(g, oracle.Options(method=0, max_iters=12, radius0=1e12, huber_delta=0.0))
    rej = [r for r in res.records if r["step_ok"] == 0]
    costs = [r["cost"] for r in res.records if r["step_ok"] == 1]
    assert all(x >= y for x, y in zip(costs, costs[1:]))  # monotone
    for r in rej:
        assert r["relative_decrease"] <= 1e-3


def test_lm_pcg_port_pose_blocks(oracle):
    """block-Jacobi over groups of B poses in the port: same LM trajectory as the direct solve, fewer PCG iterations"""
    g = oracle.read_g2o(os.path.join(DATA, "MIT.g2o"))
    a = oracle.lm_direct(g, oracle.Options(method=1, max_iters=5))
    its = {}
    for B in (1, 8, 32):
        b = oracle.lm_pcg(g, oracle.Options(method=1, max_iters=5, pcg_rtol=1e-12, pcg_max_iters=200000, threads=4,
                                             pcg_block_poses=B))
        assert np.abs(a.poses - b.poses).max() < 1e-7 and a.final_cost == pytest.approx(b.final_cost, rel=1e-9)
        its[B] = b.total_pcg_iters
    assert its[32] < its[8] < its[1]


def test_switchable_functor_restatement(oracle):
    """METHOD 2 blocks (src/ceres_error.cpp:237-317): e = s e_plain, d e / d P = s d e_plain / d P, d e / d s = e_plain,
    prior sqrt(lambda)(1 - s); checked against the plain functor and finite differences in s"""
    g = oracle.add_random_C(oracle.read_g2o(os.path.join(DATA, "INTEL.g2o")), 20, 1)
    E = g.n_edges
    rng = np.random.default_rng(3)
    sw = np.ones(E)
    sw[g.kind != 0] = rng.uniform(0.1, 1.0, int((g.kind != 0).sum()))
    c, r, J, Js, q = oracle.evaluate_sc(g, switches=sw, lam=2.0, apply_loss=False)
    c0, r0, J0 = oracle.evaluate(g, method=0, apply_loss=False)
    np.testing.assert_allclose(r, sw[:, None] * r0, rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(J, sw[:, None] * J0, rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(Js[g.kind != 0], r0[g.kind != 0], rtol=1e-14)
    assert np.all(Js[g.kind == 0] == 0) and np.all(q[g.kind == 0] == 0)
    np.testing.assert_allclose(q[g.kind != 0], np.sqrt(2.0) * (1 - sw[g.kind != 0]), rtol=1e-15)
    h = 1e-6
    k = int(np.nonzero(g.kind != 0)[0][5])
    swp, swm = sw.copy(), sw.copy()
    swp[k] += h
    swm[k] -= h
    rp = oracle.evaluate_sc(g, switches=swp, lam=2.0, apply_loss=False)[1]
    rm = oracle.evaluate_sc(g, switches=swm, lam=2.0, apply_loss=False)[1]
    np.testing.assert_allclose((rp[k] - rm[k]) / (2 * h), Js[k], rtol=1e-8)
    # Huber corrector scales r, J and Js of a block alike
    c2, r2, J2, Js2, _ = oracle.evaluate_sc(g, switches=sw, lam=2.0, apply_loss=True)
    ratio = r2[k] / r[k]
    np.testing.assert_allclose(Js2[k], ratio[0] * Js[k], rtol=1e-12)
    assert c2 == c


def test_switchable_lm_regression(oracle):
    g = oracle.add_random_C(oracle.read_g2o(os.path.join(DATA, "INTEL.g2o")), 50, 1)
    res = oracle.lm_direct_sc(g, oracle.Options(method=2, max_iters=10))
    fx = json.load(open(os.path.join(GOLDEN, "lm_INTEL_out50_m2.json")))
    for a, b in zip(res.records, fx["records"][:11]):
        assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-9)
    assert res.initial_cost == pytest.approx(3.964897979485657e+01, rel=1e-12)  # == METHOD 0 cost at s = 1


# ------------------------------------------------- optional information-weighted mode
def _closed_form_plain(P1, P2, m):
    """independent closed form of the plain SE(2) error, numpy"""
    c1, s1 = np.cos(P1[2]), np.sin(P1[2])
    D = P2[:2] - P1[:2]
    a, b = c1 * D[0] + s1 * D[1], -s1 * D[0] + c1 * D[1]
    u = np.array([a - m[0], b - m[1]])
    cd, sd = np.cos(m[2]), np.sin(m[2])
    return np.array([cd * u[0] + sd * u[1], -sd * u[0] + cd * u[1], np.arcsin(np.sin(P2[2] - P1[2] - m[2]))])


def _omega(w):
    return np.array([[w[0], w[1], w[2]], [w[1], w[3], w[4]], [w[2], w[4], w[5]]])


def test_edge_chi2_is_the_reference_mahalanobis_formula(oracle):
    """compute_edge_mahalanobis (src/layer_manager.cpp:230-282): m = r' Omega r, r plain, clamped at 0; checked against
    an independent numpy closed form on every edge of an EDGE_SE2 file and of an EDGE2 file (indefinite Omega there)"""
    fx = json.load(open(os.path.join(GOLDEN, "info_mode.json")))["chi2"]
    for name in ("INTEL", "CSAIL"):
        g = oracle.read_g2o(os.path.join(DATA, name + ".g2o"))
        got = oracle.edge_chi2(g)
        exp = np.empty(g.n_edges)
        for k in range(g.n_edges):
            r = _closed_form_plain(g.poses[g.ia[k]], g.poses[g.ib[k]], g.meas[k])
            exp[k] = max(0.0, r @ _omega(g.info[k]) @ r)
        np.testing.assert_allclose(got, exp, rtol=1e-9, atol=1e-9 * exp.max())
        assert got.sum() == pytest.approx(fx[name]["sum"], rel=1e-12)
        assert int((got == 0).sum()) == fx[name]["n_zero"]
    assert fx["CSAIL"]["n_zero"] > 10  # clamped negatives: the EDGE2 entries read positionally are not a PSD matrix


def test_info_weighting_whitens_and_uses_chi2_dcs(oracle):
    g = oracle.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    rng = np.random.default_rng(3)
    for k in (0, 500, 1230, 1400):
        P1, P2, m, w = g.poses[g.ia[k]], g.poses[g.ib[k]] + 0.05 * rng.standard_normal(3), g.meas[k], g.info[k]
        r = _closed_form_plain(P1, P2, m)
        L = np.linalg.cholesky(_omega(w))
        ew = L.T @ r
        e0 = oracle.edge(P1, P2, m, False, 1.0, False, w)
        np.testing.assert_allclose(e0, ew, rtol=1e-10, atol=1e-12)
        assert e0 @ e0 == pytest.approx(r @ _omega(w) @ r, rel=1e-10)
        chi2 = ew @ ew
        sc = min(1.0, 2.0 * 1.0 / (1.0 + chi2))
        e1, J1 = oracle.edge(P1, P2, m, True, 1.0, True, w)
        np.testing.assert_allclose(e1, sc * ew, rtol=1e-10, atol=1e-12)
        # Jacobian against central differences of the oracle's own double-precision functor
        h = 1e-6
        Jn = np.zeros((3, 6))
        for c in range(6):
            a, b, aa, bb = P1.copy(), P2.copy(), P1.copy(), P2.copy()
            (a if c < 3 else b)[c % 3] += h
            (aa if c < 3 else bb)[c % 3] -= h
            Jn[:, c] = (oracle.edge(a, b, m, True, 1.0, False, w) - oracle.edge(aa, bb, m, True, 1.0, False, w)) / (2 * h)
        assert np.abs(Jn - J1).max() < 1e-6 * max(1.0, np.abs(J1).max())
    # identity information == the unweighted METHOD 0 objective
    g2 = g.copy()
    g2.info = np.tile(np.array([1.0, 0, 0, 1.0, 0, 1.0]), (g.n_edges, 1))
    c_w, r_w, J_w = oracle.evaluate(g2, method=0, info_weighting=True)
    c_p, r_p, J_p = oracle.evaluate(g, method=0)
    assert c_w == c_p
    np.testing.assert_array_equal(r_w, r_p)
    np.testing.assert_array_equal(J_w, J_p)
    # an indefinite information matrix poisons the evaluation instead of producing numbers
    gb = oracle.read_g2o(os.path.join(DATA, "CSAIL.g2o"))
    assert np.isnan(oracle.evaluate(gb, method=0, want_r=False, want_J=False, info_weighting=True)[0])


def test_info_mode_fixture_regression(oracle):
    fx = json.load(open(os.path.join(GOLDEN, "info_mode.json")))["intel_edges_phi1"]
    g = oracle.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    for rec in fx:
        k = rec["edge"]
        for dcs in (0, 1):
            e, J = oracle.edge(g.poses[g.ia[k]], g.poses[g.ib[k]], g.meas[k], bool(dcs), 1.0, True, g.info[k])
            np.testing.assert_allclose(e, rec["e%d" % dcs], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(J.reshape(-1), rec["J%d" % dcs], rtol=1e-12, atol=1e-13)
    f = json.load(open(os.path.join(GOLDEN, "lm_MIT_out0_m0_info.json")))
    gm = oracle.read_g2o(os.path.join(DATA, "MIT.g2o"))
    res = oracle.lm_direct(gm, oracle.Options(method=0, info_weighting=1, phi=1.0))
    assert res.termination == f["termination"] and res.iterations == f["iterations"]
    assert res.final_cost == pytest.approx(f["final_cost"], rel=1e-9)
    # the C port (PCG) follows the same trajectory in this mode too
    port = oracle.lm_pcg(gm, oracle.Options(method=0, info_weighting=1, phi=1.0, max_iters=10, pcg_rtol=1e-12, pcg_block_poses=32))
    assert port.records[10]["cost"] == pytest.approx(f["records"][10]["cost"], rel=1e-6)


def test_lm_pcg_port_chain_preconditioner(oracle):
    """the chain (block-tridiagonal segments) preconditioner of the port: same LM trajectory as the 3x3 blocks when PCG
    is run tight (a preconditioner must not change what is solved), fewer PCG iterations, for several segment lengths
    including one that does not divide the pose count"""
    g = oracle.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    g = oracle.add_random_C(g, 20, 2)
    base = dict(method=1, max_iters=5, pcg_rtol=1e-12, pcg_max_iters=200000, threads=4)
    ref = oracle.lm_pcg(g, oracle.Options(pcg_block_poses=1, **base))
    for L in (8, 64, 100):
        res = oracle.lm_pcg(g, oracle.Options(pcg_chain_len=L, **base))
        assert res.total_pcg_iters < 0.5 * ref.total_pcg_iters
        for a, b in zip(res.records, ref.records):
            assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-8)
        assert np.abs(res.poses - ref.poses).max() < 1e-7
    direct = oracle.lm_direct(g, oracle.Options(method=1, max_iters=5))
    assert np.abs(ref.poses - direct.poses).max() < 1e-6
