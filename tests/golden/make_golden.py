"""Generates tests/golden/*.json from the CPU oracle (oracle/).  The reference itself cannot be run
here (needs Ceres/Eigen/Boost), so these vectors pin the ORACLE, not the reference binary: they guard
against regressions of the restatement and give the GPU tests a GPU-box-resident expected output.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")
OUT = os.path.join(ROOT, "tests", "golden")


def info_cases():
    """optional information-weighted mode (oracle/pgo_oracle.c edge_functor_jet with info): LM traces,
    per-edge chi2 (compute_edge_mahalanobis) and whitened residual/Jacobian vectors"""
    for name, n_out, method in [("INTEL", 50, 1), ("M3500", 0, 1), ("MIT", 0, 0)]:
        gg = O.read_g2o(os.path.join(DATA, name + ".g2o"))
        if n_out:
            gg = O.add_random_C(gg, n_out, 1)
        res = O.lm_direct(gg, O.Options(method=method, info_weighting=1, phi=1.0))
        tag = "%s_out%d_m%d_info" % (name, n_out, method)
        np.save(os.path.join(OUT, "lm_%s_poses.npy" % tag), res.poses)
        json.dump(dict(dataset=name, outliers=n_out, seed=1, method=method, info_weighting=1, phi=1.0,
                       termination=res.termination, iterations=res.iterations, initial_cost=res.initial_cost,
                       final_cost=res.final_cost, records=res.records), open(os.path.join(OUT, "lm_%s.json" % tag), "w"), indent=1)
        print(tag, O.TERM[res.termination], res.iterations, res.final_cost)
    chi = {}
    for name in ["INTEL", "M3500", "MIT", "CSAIL", "FR079", "FRH"]:
        gg = O.read_g2o(os.path.join(DATA, name + ".g2o"))
        c = O.edge_chi2(gg)
        chi[name] = dict(sum=float(c.sum()), max=float(c.max()), n_zero=int((c == 0).sum()), first=c[:5].tolist(), last=c[-5:].tolist())
    g = O.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    edges = []
    for k in list(range(0, 5)) + list(range(1227, 1232)):
        rec = dict(edge=int(k), info=g.info[k].tolist())
        for dcs in (0, 1):
            e, J = O.edge(g.poses[g.ia[k]], g.poses[g.ib[k]], g.meas[k], bool(dcs), 1.0, True, g.info[k])
            rec["e%d" % dcs] = e.tolist()
            rec["J%d" % dcs] = J.reshape(-1).tolist()
        edges.append(rec)
    json.dump(dict(chi2=chi, intel_edges_phi1=edges), open(os.path.join(OUT, "info_mode.json"), "w"), indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "info":
        return info_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "lm":   # only the listed LM cases: make_golden.py lm M3500_out184_m1 ...
        return lm_cases()
    info_cases()
    # per-edge residual / Jacobian vectors for 20 INTEL edges, both functors
    g = O.read_g2o(os.path.join(DATA, "INTEL.g2o"))
    idx = list(range(0, 10)) + list(range(1227, 1237))
    edges = []
    for k in idx:
        rec = dict(edge=int(k), a=int(g.ia[k]), b=int(g.ib[k]), meas=g.meas[k].tolist(), P1=g.poses[g.ia[k]].tolist(),
                   P2=g.poses[g.ib[k]].tolist())
        for dcs in (0, 1):
            e, J = O.edge(g.poses[g.ia[k]], g.poses[g.ib[k]], g.meas[k], bool(dcs))
            rec["e%d" % dcs] = e.tolist()
            rec["J%d" % dcs] = J.reshape(-1).tolist()
        edges.append(rec)
    json.dump(edges, open(os.path.join(OUT, "intel_edges.json"), "w"), indent=1)

    # initial costs, all datasets, both methods
    costs = {}
    for name in ["INTEL", "M3500", "MIT", "CSAIL", "FR079", "FRH"]:
        gg = O.read_g2o(os.path.join(DATA, name + ".g2o"))
        c0 = O.evaluate(gg, method=0, want_r=False, want_J=False)[0]
        c1, r, _ = O.evaluate(gg, method=1, apply_loss=False, want_J=False)
        c0n, r0, _ = O.evaluate(gg, method=0, apply_loss=False, want_J=False)
        n_dcs = int(np.sum(np.abs(r - r0).max(axis=1) > 0))
        costs[name] = dict(n_poses=gg.n_poses, n_edges=gg.n_edges, n_odometry=int((gg.kind == 0).sum()),
                           n_closure=int((gg.kind == 1).sum()), cost_method0=c0, cost_method1=c1, n_psi_lt_1=n_dcs)
    json.dump(costs, open(os.path.join(OUT, "initial_costs.json"), "w"), indent=1)

    # seeded bogus-edge lists on INTEL
    bog = {}
    for seed in (1, 2, 3):
        g2 = O.add_random_C(g, 50, seed)
        bog[str(seed)] = dict(a=g2.ia[-50:].tolist(), b=g2.ib[-50:].tolist(), meas_sum=float(g2.meas[-50:].sum()),
                              cost_method1=O.evaluate(g2, method=1, want_r=False, want_J=False)[0])
    json.dump(bog, open(os.path.join(OUT, "intel_bogus.json"), "w"), indent=1)

    lm_cases()


def lm_cases():
    # LM traces + final poses (direct solve) for the BASELINE configs C1..C3
    cases = [("INTEL", 50, 1), ("INTEL", 50, 0), ("INTEL", 0, 1), ("INTEL", 0, 0), ("MIT", 0, 1), ("MIT", 0, 0), ("M3500", 0, 1),
             ("M3500", 0, 0), ("CSAIL", 0, 1), ("FR079", 0, 1), ("FRH", 0, 1), ("FRH", 20, 1)]
    # "0 and 10 %-of-closures bogus edges, METHOD 0 and 1" on M3500 (1844 closures) and MIT (20 closures)
    cases += [("M3500", 184, 1), ("M3500", 184, 0), ("MIT", 2, 1), ("MIT", 2, 0)]
    only = set(sys.argv[2:]) if len(sys.argv) > 2 and sys.argv[1] == "lm" else None
    for name, n_out, method in cases:
        if only is not None and "%s_out%d_m%d" % (name, n_out, method) not in only:
            continue
        gg = O.read_g2o(os.path.join(DATA, name + ".g2o"))
        if n_out:
            gg = O.add_random_C(gg, n_out, 1)
        res = O.lm_direct(gg, O.Options(method=method))
        tag = "%s_out%d_m%d" % (name, n_out, method)
        np.save(os.path.join(OUT, "lm_%s_poses.npy" % tag), res.poses)
        json.dump(dict(dataset=name, outliers=n_out, seed=1, method=method, termination=res.termination,
                       iterations=res.iterations, initial_cost=res.initial_cost, final_cost=res.final_cost,
                       records=res.records), open(os.path.join(OUT, "lm_%s.json" % tag), "w"), indent=1)
        print(tag, O.TERM[res.termination], res.iterations, res.final_cost)

    if only is not None:
        return
    # METHOD 2 (switchable constraints): joint LM over poses and switches
    for name, n_out in [("INTEL", 50), ("M3500", 0), ("MIT", 0)]:
        gg = O.read_g2o(os.path.join(DATA, name + ".g2o"))
        if n_out:
            gg = O.add_random_C(gg, n_out, 1)
        res = O.lm_direct_sc(gg, O.Options(method=2))
        tag = "%s_out%d_m2" % (name, n_out)
        np.save(os.path.join(OUT, "lm_%s_poses.npy" % tag), res.poses)
        np.save(os.path.join(OUT, "lm_%s_switches.npy" % tag), res.switches)
        json.dump(dict(dataset=name, outliers=n_out, seed=1, method=2, termination=res.termination,
                       iterations=res.iterations, initial_cost=res.initial_cost, final_cost=res.final_cost,
                       records=res.records), open(os.path.join(OUT, "lm_%s.json" % tag), "w"), indent=1)
        print(tag, O.TERM[res.termination], res.iterations, res.final_cost)


if __name__ == "__main__":
    main()
