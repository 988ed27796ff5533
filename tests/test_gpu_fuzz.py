"""GPU: property-based tests over small random pose graphs (hypothesis): the edges of the hot path's input space that the
datasets do not reach -- duplicate pairs, edges with a > b, missing odometry edges, several connected components, a
constant pose other than pose 0 (or none), hubs with hundreds of incident edges, 2 .. 600 poses.  Every example runs
K1 / K2 / K3 and one or two LM iterations through the C-ABI against the CPU oracle, and checks that the direct solve is
either taken or refused with PGO_ERR_UNSUPPORTED -- never silently replaced.

The examples are derived deterministically (derandomize) so that a failure reproduces; sizes keep the whole file within
about a minute on the GPU box."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, note, settings, strategies as st

from conftest import DATA, GOLDEN, oracle_graph

pytestmark = pytest.mark.gpu

# PGO_FUZZ_EXAMPLES=N: a longer, randomised campaign (scripts/exp_fuzz.sh); the suite's default is 30 / 20 derandomised examples
FUZZ_N = int(os.environ.get("PGO_FUZZ_EXAMPLES", "0"))
SETTINGS = dict(max_examples=FUZZ_N or 30, deadline=None, derandomize=not FUZZ_N, suppress_health_check=list(HealthCheck))


def make_graph(n, seed, p_chain, n_extra, hub, dup):
    """the graph of one recipe (plain function: scripts/fuzz_case.py replays a failing recipe with it)"""
    rng = np.random.default_rng(seed)
    poses = np.column_stack([rng.uniform(-8, 8, n), rng.uniform(-8, 8, n), rng.uniform(-3, 3, n)])
    ia, ib, kind = [], [], []
    for i in range(n - 1):
        if rng.uniform() < p_chain:
            a, b = (i, i + 1) if rng.uniform() < 0.8 else (i + 1, i)    # edges with a > b
            ia.append(a); ib.append(b); kind.append(0)
    for _ in range(n_extra):
        a, b = int(rng.integers(n)), int(rng.integers(n))
        if a == b:
            continue
        ia.append(a); ib.append(b); kind.append(int(rng.integers(1, 3)))
    if hub:
        h = int(rng.integers(n))
        for leaf in rng.choice(n, size=min(n - 1, int(rng.integers(10, 320))), replace=False):
            if int(leaf) != h:
                ia.append(h if rng.uniform() < 0.5 else int(leaf)); ib.append(int(leaf) if ia[-1] == h else h); kind.append(1)
    for _ in range(dup):
        if ia:
            k = int(rng.integers(len(ia)))
            ia.append(ia[k]); ib.append(ib[k]); kind.append(kind[k] if kind[k] else 1)   # a duplicated pair
    ia, ib = np.array(ia, np.int32).reshape(-1), np.array(ib, np.int32).reshape(-1)
    kind = np.array(kind, np.uint8).reshape(-1)
    # heading errors within +-1.2 rad: d asin(sin delta) is ill-conditioned next to |sin delta| = 1 (covered by
    # test_edge_kernel_special_cases); translation errors up to a few metres so that DCS is active on some edges
    dth = poses[ib, 2] - poses[ia, 2] - rng.uniform(-1.2, 1.2, len(ia)) if len(ia) else np.zeros(0)
    meas = np.column_stack([rng.uniform(-2, 2, len(ia)), rng.uniform(-2, 2, len(ia)), dth]) if len(ia) else np.zeros((0, 3))
    return poses, ia, ib, meas, kind


@st.composite
def pose_graphs(draw):
    n = draw(st.integers(2, 600))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    p_chain = draw(st.sampled_from([1.0, 1.0, 0.9, 0.5, 0.0]))        # missing odometry edges -> several components
    n_extra = draw(st.integers(0, 3 * n))
    hub = draw(st.booleans()) and n > 8
    dup = draw(st.integers(0, 5))
    fixed = draw(st.sampled_from([0, 0, n - 1, n // 2, -1]))
    method = draw(st.sampled_from([0, 1, 1, 2]))
    note("pose_graphs recipe: n=%d seed=%d p_chain=%g n_extra=%d hub=%s dup=%d fixed=%d method=%d" % (n, seed, p_chain, n_extra, hub, dup, fixed, method))
    return make_graph(n, seed, p_chain, n_extra, hub, dup) + (fixed, method)


class _layout:
    """0: the library's choice; 1: the large-graph layout (every row tile in 256 incidence slots of its own, null incidences behind
    the real ones) with its product kernel k_spmv_1 and the folded dot partials, forced on a small graph through the test hooks;
    2: that layout under the small-graph PCG loop (direction update inside the product, k_spmv_t MODE 5)"""
    def __init__(self, pgo, mode):
        self.pgo, self.mode = pgo, mode

    def __enter__(self):
        if self.mode:
            self.pgo.set_knob("pad_tiles", 1)
        if self.mode == 1:
            self.pgo.set_knob("fused_p", 0)

    def __exit__(self, *exc):
        self.pgo.set_knob("pad_tiles", -1)
        self.pgo.set_knob("fused_p", -1)


@settings(**SETTINGS)
@given(pose_graphs(), st.sampled_from([0, 1, 1, 2]))
def test_kernels_and_one_lm_iteration_against_the_oracle(pgo, oracle, case, layout):
    poses, ia, ib, meas, kind, fixed, method = case
    g = pgo.Graph.from_arrays(poses, ia, ib, meas, kind)
    og = oracle_graph(oracle, g)
    m12 = method if method != 2 else 1        # (the oracle's METHOD 2 evaluator has its own entry point: kernels are compared on 0 / 1)
    with _layout(pgo, layout):
        s = pgo.Solver(g, pgo.Options(method=m12, fixed_pose=fixed, linear_solver=1))
    # K1
    c, r, J = s.evaluate()
    oc, orr, oJ = oracle.evaluate(og, method=m12)
    assert c == pytest.approx(oc, rel=1e-12, abs=1e-300)
    if g.n_edges:
        assert np.abs(r - orr).max() < 1e-11 and np.abs(J - oJ).max() < 1e-10
    # K2 / K3
    grad, hd = s.normal_eq()
    x = np.random.default_rng(1).standard_normal(3 * g.n_poses)
    ograd, ohd, oy = oracle.normal_eq(og, method=m12, x=x, fixed_pose=fixed)
    sc = max(1.0, np.abs(ohd).max())
    assert np.abs(grad - ograd).max() < 1e-11 * sc and np.abs(hd - ohd).max() < 1e-11 * sc
    assert np.abs(s.spmv(x) - oy).max() < 1e-10 * sc
    s.close()
    # two LM iterations, tight PCG, against the C port of the same policy (METHOD 0 / 1)
    if method != 2 and g.n_edges:
        kw = dict(method=method, fixed_pose=fixed, max_iters=2, pcg_rtol=1e-12, pcg_max_iters=100000)
        with _layout(pgo, layout):
            s2 = pgo.Solver(g, pgo.Options(linear_solver=1, **kw))
        summ = s2.solve()
        ores = oracle.lm_pcg(og, oracle.Options(**kw))
        assert [a["step_ok"] for a in s2.iter_records()] == [b["step_ok"] for b in ores.records]
        assert summ.final_cost == pytest.approx(ores.final_cost, rel=1e-7, abs=1e-12)
        assert np.abs(s2.poses() - ores.poses).max() < 1e-6 * max(1.0, np.abs(ores.poses).max())
        if fixed >= 0:
            np.testing.assert_array_equal(s2.poses()[fixed], poses[fixed])
        s2.close()


@settings(**dict(SETTINGS, max_examples=FUZZ_N or 20))
@given(pose_graphs())
def test_direct_solve_is_taken_or_refused_never_replaced(pgo, case):
    """linear_solver = 2 on an arbitrary graph: either the handle IS on the direct solve and its LM iterations agree with
    PCG to 1e-10, or pgo_create fails with PGO_ERR_UNSUPPORTED (missing chain edge, no constant pose, too many edges
    outside the chain ...) -- a silent fallback to PCG would hide a wrong answer about which solver ran"""
    poses, ia, ib, meas, kind, fixed, method = case
    g = pgo.Graph.from_arrays(poses, ia, ib, meas, kind)
    kw = dict(method=method, fixed_pose=fixed, max_iters=2, pcg_rtol=1e-12, pcg_max_iters=100000)
    try:
        s = pgo.Solver(g, pgo.Options(linear_solver=2, **kw))
    except pgo.PgoError as e:
        assert e.status == -8, str(e)     # PGO_ERR_UNSUPPORTED
        n = len(poses)
        pairs = set(zip(np.minimum(ia, ib).tolist(), np.maximum(ia, ib).tolist()))
        chain_ok = all((i, i + 1) in pairs for i in range(n - 1))
        outside = len(ia) - (n - 1)
        assert (not chain_ok) or fixed < 0 or 3 * outside + 1 > 6144 or n < 2, "refused although the graph qualifies"
        return
    assert s.info().linear_solver == 2
    sm = s.solve()
    ref = pgo.Solver(g, pgo.Options(linear_solver=1, pcg_coarse_poses=0, **kw))
    sr = ref.solve()
    assert s.info().linear_solver == 2 and all(r["pcg_iters"] == 0 for r in s.iter_records()) or s.info().direct_fallbacks > 0
    assert [a["step_ok"] for a in s.iter_records()] == [b["step_ok"] for b in ref.iter_records()]
    assert sm.final_cost == pytest.approx(sr.final_cost, rel=1e-7, abs=1e-12)
    assert np.abs(s.poses() - ref.poses()).max() < 1e-6 * max(1.0, np.abs(ref.poses()).max())
    s.close(); ref.close()


@settings(**dict(SETTINGS, max_examples=FUZZ_N or 20))
@given(pose_graphs(), st.sampled_from([3, 8, 16, 50]))
def test_coarse_level_and_single_reduction_loop_on_arbitrary_graphs(pgo, case, agg):
    """the round-3 solver paths on graphs they were not tuned on (several components, edge-less poses, hubs, a constant pose
    anywhere or none, aggregates that do not divide anything): a tight solve with the second preconditioner level and a
    solve with the one-reduction PCG loop must both land where the plain one-level, two-reduction solve lands -- a
    preconditioner or a reformulated recurrence may change the iteration count, never the answer"""
    poses, ia, ib, meas, kind, fixed, method = case
    if len(ia) == 0:
        return
    g = pgo.Graph.from_arrays(poses, ia, ib, meas, kind)
    kw = dict(method=method, fixed_pose=fixed, max_iters=3, pcg_rtol=1e-12, pcg_max_iters=200000, linear_solver=1)
    ref = pgo.Solver(g, pgo.Options(pcg_coarse_poses=0, **kw))
    sr = ref.solve()
    two = pgo.Solver(g, pgo.Options(pcg_coarse_poses=agg, **kw))
    s2 = two.solve()
    assert two.info().pcg_coarse_poses == agg and two.info().pcg_coarse_rank == 3 * ((len(poses) + agg - 1) // agg)
    assert [r["step_ok"] for r in two.iter_records()] == [r["step_ok"] for r in ref.iter_records()]
    c0 = ref.iter_records()[0]["cost"]      # (final costs next to zero -- consistent measurements -- are compared on the scale of the initial cost)
    assert s2.final_cost == pytest.approx(sr.final_cost, rel=1e-7, abs=1e-12 + 1e-8 * c0)
    assert np.abs(two.poses() - ref.poses()).max() < 1e-6 * max(1.0, np.abs(ref.poses()).max())
    assert all(r["iter"] == 0 or r["pcg_rel_residual"] <= 1e-12 for r in two.iter_records())
    if fixed >= 0:
        np.testing.assert_array_equal(two.poses()[fixed], poses[fixed])
    tight_cost, tight_poses = sr.final_cost, ref.poses().copy()
    ref.close(); two.close()
    # the one-reduction loop needs the chain preconditioner and the inexact mode; one rank through the test hook
    if len(poses) >= 8 and method != 2:
        kw2 = dict(method=method, fixed_pose=fixed, max_iters=3, pcg_rtol=1e-6, pcg_max_iters=200000, linear_solver=1, pcg_chain_len=8,
                   pcg_coarse_poses=0)
        # "verify_residual": the records carry the TRUE residual |b - A y| / |b| of every PCG solve, not the recurrence
        # residual the loop stopped on
        pgo.set_knob("verify_residual", 1)
        pgo.set_knob("fused_p", 0)
        try:
            a = pgo.Solver(g, pgo.Options(**kw2))
            pgo.set_knob("single_reduction", 1)
            b = pgo.Solver(g, pgo.Options(**kw2))
        finally:
            for k in ("single_reduction", "fused_p", "verify_residual"):
                pgo.set_knob(k, -1)
        sa, sb = a.solve(), b.solve()
        assert b.info().pcg_single_reduction == 1 and a.info().pcg_single_reduction == 0
        # Both loops stop at |r| <= 1e-6 |b| of their RECURRENCE residuals.  On the ill-conditioned LM systems of tiny random
        # graphs that leaves the step open by far more than 1e-6 -- replayed recipes (scripts/fuzz_case.py): the textbook loop
        # ends 1.6e-3 .. 2.0 from the tight solve's poses, the one-reduction loop 3.8e-3 .. 1.7, neither monotone in the
        # tolerance -- so the two answers are not compared with each other.  What each loop must deliver is its own claim: a
        # solution whose TRUE residual is of the order of the tolerance; the reformulated recurrences may drift from it
        # more than the textbook ones, but not by an order of magnitude.
        assert [r["step_ok"] for r in a.iter_records()][:2] == [r["step_ok"] for r in b.iter_records()][:2]
        ra, rb = a.iter_records()[1], b.iter_records()[1]        # the first LM iteration: the same linear system for both
        assert abs(ra["pcg_iters"] - rb["pcg_iters"]) <= 3 + 0.3 * ra["pcg_iters"]
        assert rb["pcg_rel_residual"] <= 10.0 * max(ra["pcg_rel_residual"], 1e-6)
        for r in b.iter_records()[1:]:
            assert r["pcg_rel_residual"] <= 1e-4          # (every solve of the one-reduction loop: within 100 x its tolerance)
        # and the three LM iterations land near the tight solve on the scale of what they started from
        c0 = a.iter_records()[0]["cost"]
        assert abs(sb.final_cost - tight_cost) <= 10.0 * abs(sa.final_cost - tight_cost) + 2e-2 * c0
        a.close(); b.close()


@st.composite
def graph_bundles(draw):
    k = draw(st.integers(1, 9))
    recipes = []
    for _ in range(k):
        n = draw(st.sampled_from([2, 3, 7, 64, 65, 85, 86, 200, 255, 256, 257, 300]) | st.integers(2, 400))
        recipes.append((n, draw(st.integers(0, 2 ** 31 - 1)), draw(st.sampled_from([1.0, 1.0, 0.7])), draw(st.integers(0, 2 * n)),
                        draw(st.booleans()) and n > 8, draw(st.integers(0, 3))))
    method = draw(st.sampled_from([0, 1]))
    fixed = draw(st.sampled_from([0, 0, -1]))
    note("graph_bundles recipe: %r method=%d fixed=%d" % (recipes, method, fixed))
    return recipes, method, fixed


@settings(**dict(SETTINGS, max_examples=FUZZ_N or 15))
@given(graph_bundles())
def test_batched_handle_on_arbitrary_bundles(pgo, bundle):
    """pgo_batch_*: bundles of 1-9 arbitrary small graphs (row counts around the tile limits 85 / 256, edge-less poses,
    hubs, duplicate pairs, with and without a constant pose) in ONE handle -- every problem must come out as its own
    single-handle PCG solve does: same accept / reject history, cost to 1e-9 of the initial cost, poses to 1e-7"""
    recipes, method, fixed = bundle
    graphs = []
    for rc in recipes:
        poses, ia, ib, meas, kind = make_graph(*rc)
        if len(ia) == 0:
            return
        graphs.append(pgo.Graph.from_arrays(poses, ia, ib, meas, kind))
    opt = dict(method=method, fixed_pose=fixed, max_iters=3, pcg_rtol=1e-12, pcg_max_iters=100000)
    max_deg = max(int(np.bincount(np.concatenate([np.array(g.ia), np.array(g.ib)])).max()) for g in graphs)
    try:
        b = pgo.Batch(graphs, pgo.Options(**opt))
    except pgo.PgoError as e:        # the one documented refusal: a pose with more incident edges than a row tile holds
        assert e.status == -8 and max_deg > 256, str(e)
        return
    assert max_deg <= 256
    sb = b.solve()
    for k, g in enumerate(graphs):
        s = pgo.Solver(g, pgo.Options(linear_solver=1, pcg_coarse_poses=0, **opt))
        ss = s.solve()
        c0 = max(ss.initial_cost, 1e-30)
        assert sb[k].initial_cost == pytest.approx(ss.initial_cost, rel=1e-12, abs=1e-300)
        assert [r["step_ok"] for r in b.iter_records(k)] == [r["step_ok"] for r in s.iter_records()], k
        assert abs(sb[k].final_cost - ss.final_cost) <= 1e-9 * c0, k
        assert np.abs(b.poses(k) - s.poses()).max() < 1e-7 * max(1.0, np.abs(s.poses()).max()), k
        s.close()
    b.close()


def test_mit_distance_from_the_fixture_is_the_conditioning(pgo):
    """MIT METHOD 1 runs at a trust-region radius of ~3e11, where the LM systems are so ill-conditioned that ANY two accurate
    solves end ~1e-6 apart after 50 iterations: two independent HIP solves -- PCG to 1e-13 and the direct chain + low-rank
    solve with refinement -- differ from each other (measured 5e-7) by about as much as each differs from the golden fixture
    (the oracle's SuperLU LM solve; 1.0e-6 / 1.5e-6): three distances of one order.  All far inside north_star's 1e-4."""
    g = pgo.ReadG2O(os.path.join(DATA, "MIT.g2o"))
    ref = np.load(os.path.join(GOLDEN, "lm_MIT_out0_m1_poses.npy"))
    out = {}
    for name, kw in (("pcg", dict(linear_solver=1, pcg_rtol=1e-13, pcg_max_iters=2000000, pcg_coarse_poses=0)), ("direct", dict(linear_solver=2))):
        s = pgo.Solver(g, pgo.Options(method=1, **kw))
        s.solve()
        out[name] = s.poses()
        s.close()
    d_pd = np.abs(out["pcg"][:, :2] - out["direct"][:, :2]).max()
    d_pf = np.abs(out["pcg"][:, :2] - ref[:, :2]).max()
    d_df = np.abs(out["direct"][:, :2] - ref[:, :2]).max()
    print("MIT METHOD 1, 50 LM iterations: |PCG(1e-13) - direct| %.2e, |PCG - fixture| %.2e, |direct - fixture| %.2e" % (d_pd, d_pf, d_df))
    assert max(d_pd, d_pf, d_df) < 1e-5          # an order of magnitude inside north_star's tolerance, whichever pair is compared
    assert d_pd > 1e-9                            # ... and the two HIP solves do differ at that level: it is the conditioning,
    assert max(d_pd, d_pf, d_df) < 20.0 * min(d_pd, d_pf, d_df)   # not one solve (or the fixture) being off
