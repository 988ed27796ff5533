"""GPU: the pose-id-range sharded solve (one process per rank) against the single-rank solve.

The production communicator is RCCL (one GPU per rank).  On a one-GPU box RCCL refuses several ranks
on the same device, so these tests drive the identical solver code through the host-staged
shared-memory communicator (pgo_comm_create_shm): same shards, same collectives (all-reduce of the CG
dot products and LM scalars, all-gather of the search direction / poses), ranks as separate processes
sharing cuda:0."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "_shard_worker.py")


_RUNS = [0]


def run(world, cfg, tmp, env=None, tag=""):
    out = os.path.join(str(tmp), "w%d%s" % (world, tag))
    os.makedirs(out, exist_ok=True)
    _RUNS[0] += 1
    name = "pgo_test_%d_%d_%d" % (os.getpid(), world, _RUNS[0])    # (a fresh segment per run: a failed run leaves its own behind)
    procs = []
    for r in range(world):
        c = dict(cfg, rank=r, world=world, name=name, out=out)
        procs.append(subprocess.Popen([sys.executable, WORKER, json.dumps(c)], stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **(env or {}))))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    if any(p.returncode != 0 for p in procs):   # every rank's log: the first failing rank often only reports that a peer went away
        raise AssertionError("\n".join("rank %d exit %s:\n%s" % (r, p.returncode, logs[r][-3000:]) for r, p in enumerate(procs)))
    res = [json.load(open(os.path.join(out, "out_%d.json" % r))) for r in range(world)]
    poses = [np.load(os.path.join(out, "poses_%d.npy" % r)) for r in range(world)]
    return res, poses


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tight_solve_intel(tmp_path, world):
    cfg = dict(graph="INTEL", outliers=50, options=dict(method=1, max_iters=4, pcg_rtol=1e-11, pcg_max_iters=30000))
    ref, ref_poses = run(1, cfg, tmp_path)
    res, poses = run(world, cfg, tmp_path)
    for r in range(world):
        # every rank holds the full, identical pose vector and the same LM history
        np.testing.assert_array_equal(poses[r], poses[0])
        assert res[r]["cost0"] == pytest.approx(ref[0]["cost0"], rel=1e-13)
        assert res[r]["summary"]["iterations"] == 4
        for a, b in zip(res[r]["records"], ref[0]["records"]):
            assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-9)
    assert np.abs(poses[0] - ref_poses[0]).max() < 1e-7


@pytest.mark.parametrize("world,halo,chain", [(2, 0, 0), (4, 0, 0), (2, 1, 0), (3, 1, 0), (4, 1, 0), (2, 1, 64), (3, 0, 64), (4, 1, 64)])
def test_sharded_inexact_solve_synthetic(tmp_path, world, halo, chain):
    """halo = 0: all-gather of the search direction; halo = 1: point-to-point exchange of the referenced rows;
    chain = 64: the chain preconditioner (shards aligned to its segments) instead of the dense 4-pose blocks"""
    cfg = dict(graph="synth", n_poses=20001, seed=9,
               options=dict(method=1, max_iters=4, pcg_rtol=0.1, pcg_max_iters=300, pcg_chain_len=chain))
    ref, ref_poses = run(1, cfg, tmp_path, tag="c%d" % chain)
    cfg = dict(cfg, options=dict(cfg["options"], halo_exchange=halo))
    res, poses = run(world, cfg, tmp_path, tag="h%dc%d" % (halo, chain))
    for r in range(world):
        np.testing.assert_array_equal(poses[r], poses[0])
        for a, b in zip(res[r]["records"], ref[0]["records"]):
            assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-8)
            assert abs(a["pcg_iters"] - b["pcg_iters"]) <= 1  # shard boundaries are aligned to the preconditioner blocks
    assert np.abs(poses[0] - ref_poses[0]).max() < 1e-6


def test_rccl_backend_single_rank(tmp_path):
    """RCCL communicator with the collectives forced on at world == 1 (identities): exercises
    ncclCommInitRank / ncclAllReduce / in-place ncclAllGather exactly as the multi-GPU solve issues them."""
    cfg = dict(graph="INTEL", outliers=50, options=dict(method=1, max_iters=3, pcg_rtol=1e-10, pcg_max_iters=30000, linear_solver=1))
    ref, ref_poses = run(1, cfg, tmp_path)
    res, poses = run(1, dict(cfg, comm="rccl"), tmp_path, env={"PGO_FORCE_COLLECTIVES": "1"}, tag="rccl")
    assert res[0]["summary"]["iterations"] == 3
    assert res[0]["summary"]["final_cost"] == pytest.approx(ref[0]["summary"]["final_cost"], rel=1e-10)
    assert np.abs(poses[0] - ref_poses[0]).max() < 1e-8


@pytest.mark.parametrize("world,halo", [(2, 0), (3, 1)])
def test_sharded_switchable_constraints(tmp_path, world, halo):
    """METHOD 2 across ranks: cut edges keep a replica of their switch on both owners and update it identically"""
    cfg = dict(graph="INTEL", outliers=50, options=dict(method=2, max_iters=4, pcg_rtol=1e-11, pcg_max_iters=30000))
    ref, ref_poses = run(1, cfg, tmp_path)
    cfg = dict(cfg, options=dict(cfg["options"], halo_exchange=halo))
    res, poses = run(world, cfg, tmp_path, tag="sc%d" % halo)
    for r in range(world):
        np.testing.assert_array_equal(poses[r], poses[0])
        for a, b in zip(res[r]["records"], ref[0]["records"]):
            assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-9)
            assert a["gradient_max_norm"] == pytest.approx(b["gradient_max_norm"], rel=1e-7)
    assert np.abs(poses[0] - ref_poses[0]).max() < 1e-7


@pytest.mark.parametrize("world,halo", [(2, 1), (3, 0)])
def test_sharded_info_weighting_and_chi2(tmp_path, world, halo):
    """information-weighted mode (128-byte records) and pgo_edge_chi2 across ranks: cut edges are whitened on both
    owners; the chi2 vector is assembled by one all-reduce and is identical on every rank"""
    cfg = dict(graph="M3500", chi2=1,
               options=dict(method=1, info_weighting=1, phi=1.0, max_iters=4, pcg_rtol=1e-11, pcg_max_iters=60000))
    ref, ref_poses = run(1, cfg, tmp_path)
    ref_chi2 = np.load(os.path.join(str(tmp_path), "w1", "chi2_0.npy"))
    cfg = dict(cfg, options=dict(cfg["options"], halo_exchange=halo))
    res, poses = run(world, cfg, tmp_path, tag="info%d" % halo)
    for r in range(world):
        np.testing.assert_array_equal(poses[r], poses[0])
        for a, b in zip(res[r]["records"], ref[0]["records"]):
            assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-9)
        chi2 = np.load(os.path.join(str(tmp_path), "w%dinfo%d" % (world, halo), "chi2_%d.npy" % r))
        np.testing.assert_allclose(chi2, ref_chi2, rtol=1e-6, atol=1e-9)  # evaluated at the solved poses (1e-7 apart)
    assert np.abs(poses[0] - ref_poses[0]).max() < 1e-7


@pytest.mark.parametrize("world,chain", [(2, 64), (4, 0)])
def test_halo_overlap_matches_plain_exchange(tmp_path, world, chain):
    """halo_overlap = 1 (opt-in): the blocks with owned columns are multiplied (k_spmv MODE 4) while the exchange runs on
    a second stream, the blocks with remote columns afterwards (k_spmv_remote); 0: exchange, then one SpMV launch.
    Same LM history either way; also against the 1-rank solve."""
    base = dict(graph="synth", n_poses=30001, seed=5,
                options=dict(method=1, max_iters=4, pcg_rtol=0.1, pcg_max_iters=300, pcg_chain_len=chain, halo_exchange=1))
    ref, ref_poses = run(1, base, tmp_path, tag="ref%d" % chain)
    out = {}
    for ov in (0, 1):
        cfg = dict(base, options=dict(base["options"], halo_overlap=ov))
        out[ov] = run(world, cfg, tmp_path, tag="ov%dc%d" % (ov, chain))
    for ov in (0, 1):
        res, poses = out[ov]
        for r in range(world):
            np.testing.assert_array_equal(poses[r], poses[0])
            for a, b in zip(res[r]["records"], ref[0]["records"]):
                assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-8)
                assert abs(a["pcg_iters"] - b["pcg_iters"]) <= 1
        assert np.abs(poses[0] - ref_poses[0]).max() < 1e-6
    assert np.abs(out[0][1][0] - out[1][1][0]).max() < 1e-9


@pytest.mark.parametrize("world,halo,rtol", [(2, 0, 0.1), (4, 1, 0.1), (3, 1, 1e-3), (2, 0, 1e-6)])
def test_single_reduction_pcg_matches_two_reduction_loop(tmp_path, world, halo, rtol):
    """several ranks, inexact mode (pcg_rtol >= 1e-6), chain preconditioner: the PCG loop with ONE reduction point per
    iteration (k_cg_sr_*: Chronopoulos-Gear recurrences, (gamma, rr, delta) in one all-reduce) is the default; the same
    solve with the textbook two-reduction loop (test hook) and on one rank gives the same LM history, PCG iteration counts
    within 2 per LM iteration, and the same poses"""
    base = dict(graph="synth", n_poses=30001, seed=11,
                options=dict(method=1, max_iters=5, pcg_rtol=rtol, pcg_max_iters=20000, pcg_chain_len=64, halo_exchange=halo))
    ref, ref_poses = run(1, base, tmp_path, tag="ref")
    two, two_poses = run(world, dict(base, knobs=dict(single_reduction=0)), tmp_path, tag="two")
    one, one_poses = run(world, base, tmp_path, tag="one")
    assert ref[0]["info"]["pcg_single_reduction"] == 0 and two[0]["info"]["pcg_single_reduction"] == 0
    for r in range(world):
        assert one[r]["info"]["pcg_single_reduction"] == 1
        np.testing.assert_array_equal(one_poses[r], one_poses[0])
        for a, b, c in zip(one[r]["records"], two[r]["records"], ref[0]["records"]):
            assert a["step_ok"] == b["step_ok"] == c["step_ok"]
            assert a["cost"] == pytest.approx(b["cost"], rel=1e-8) and a["cost"] == pytest.approx(c["cost"], rel=1e-8)
            assert abs(a["pcg_iters"] - b["pcg_iters"]) <= 2 and abs(a["pcg_iters"] - c["pcg_iters"]) <= 2
            assert a["iter"] == 0 or a["pcg_rel_residual"] <= rtol
    print("world %d rtol %g: PCG iterations one-reduction %d, two-reduction %d, one rank %d; host enqueue %.1f / %.1f us per PCG iteration"
          % (world, rtol, one[0]["summary"]["total_pcg_iters"], two[0]["summary"]["total_pcg_iters"], ref[0]["summary"]["total_pcg_iters"],
             one[0]["info"]["host_enqueue_us_per_pcg_iter"], two[0]["info"]["host_enqueue_us_per_pcg_iter"]))
    # (five inexact LM iterations: the iterates of two correct loops drift apart by rounding, amplified by the loose solves)
    assert np.abs(one_poses[0] - two_poses[0]).max() < 1e-5 and np.abs(one_poses[0] - ref_poses[0]).max() < 1e-5


@pytest.mark.parametrize("n,world,chain,halo,rtol", [(63, 2, 64, 1, 1e-3), (65, 4, 8, 1, 1e-12), (65, 4, 8, 0, 1e-3), (3, 4, 0, 1, 1e-12),
                                                     (9, 3, 8, 1, 1e-12), (3, 3, 8, 0, 1e-3)])
def test_ranks_that_own_no_rows(tmp_path, n, world, chain, halo, rtol):
    """more ranks than the alignment leaves shards for (63 poses in 64-pose chain segments on 2 ranks: rank 1 owns nothing):
    every rank must still resolve to the same PCG loop and issue the same collectives -- found by scripts/exp_shard_fuzz.py,
    where the empty rank kept the two-reduction loop while its peers ran the one-reduction loop"""
    cfg = dict(graph="recipe", recipe=[n, 1234 + n, 1.0, n, False, 1], knobs=dict(shm_timeout_s=20),
               options=dict(method=1, fixed_pose=0, max_iters=3, pcg_rtol=rtol, pcg_max_iters=100000, linear_solver=1, halo_exchange=halo,
                            pcg_chain_len=chain))
    ref, ref_poses = run(1, cfg, tmp_path)
    res, poses = run(world, cfg, tmp_path)
    for r in range(world):
        np.testing.assert_array_equal(poses[r], poses[0])
        assert res[r]["info"]["pcg_single_reduction"] == res[0]["info"]["pcg_single_reduction"]
        assert [a["step_ok"] for a in res[r]["records"]] == [b["step_ok"] for b in ref[0]["records"]]
    if rtol < 1e-6:
        assert res[0]["summary"]["final_cost"] == pytest.approx(ref[0]["summary"]["final_cost"], rel=1e-8, abs=1e-12)
        assert np.abs(poses[0] - ref_poses[0]).max() < 1e-6 * max(1.0, np.abs(ref_poses[0]).max())


def test_sharded_solve_with_one_tile_product_kernel(tmp_path):
    """shards of more than 4096 row tiles: every rank multiplies with k_spmv_1 (one tile per workgroup, one dot partial
    per tile) and reduce_to_scal folds those partials (k_fold_partials) in front of k_finalize / the all-reduce -- in the
    one-reduction loop (default) and in the two-reduction loop; same LM history as one rank"""
    base = dict(graph="synth", n_poses=300001, seed=21, options=dict(method=1, max_iters=3, pcg_rtol=0.1, pcg_max_iters=500))
    ref, ref_poses = run(1, base, tmp_path, tag="ref")
    for knobs, tag in ((None, "one"), (dict(single_reduction=0), "two")):
        res, poses = run(2, dict(base, knobs=knobs), tmp_path, tag=tag)
        for r in range(2):
            assert res[r]["info"]["n_tiles"] > 4096 and res[r]["info"]["pcg_single_reduction"] == (knobs is None)
            np.testing.assert_array_equal(poses[r], poses[0])
            for a, b in zip(res[r]["records"], ref[0]["records"]):
                assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-8)
                assert abs(a["pcg_iters"] - b["pcg_iters"]) <= 2
        assert np.abs(poses[0] - ref_poses[0]).max() < 1e-5


def test_single_reduction_pcg_in_a_captured_graph_with_rccl(tmp_path):
    """world == 1 through RCCL with the collectives forced on: the one-reduction loop -- kernels, all-gather, all-reduce --
    is captured into the PCG hipGraph and replayed; same result as the plain single-rank solve.  The exact mode
    (pcg_rtol < 1e-6) keeps the two-reduction loop."""
    base = dict(graph="synth", n_poses=60001, seed=4, options=dict(method=1, max_iters=4, pcg_rtol=0.1, pcg_max_iters=500))
    ref, ref_poses = run(1, base, tmp_path, tag="ref")
    res, poses = run(1, dict(base, comm="rccl"), tmp_path, env={"PGO_FORCE_COLLECTIVES": "1"}, tag="rccl")
    assert res[0]["info"]["pcg_single_reduction"] == 1 and res[0]["info"]["pcg_graph_replay"] == 1
    for a, b in zip(res[0]["records"], ref[0]["records"]):
        assert a["step_ok"] == b["step_ok"] and a["cost"] == pytest.approx(b["cost"], rel=1e-8) and abs(a["pcg_iters"] - b["pcg_iters"]) <= 2
    assert np.abs(poses[0] - ref_poses[0]).max() < 1e-6
    tight, _ = run(1, dict(base, comm="rccl", options=dict(base["options"], pcg_rtol=1e-9, pcg_max_iters=50000, max_iters=2)), tmp_path,
                   env={"PGO_FORCE_COLLECTIVES": "1"}, tag="tight")
    assert tight[0]["info"]["pcg_single_reduction"] == 0
