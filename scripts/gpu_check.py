"""Ad-hoc GPU check used while bringing the backend up (the real tests live in tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
import toy_robust_backend_slam_amd as P

def og(g):
    return O.Graph(np.array(g.pose_ids), np.array(g.poses), np.array(g.ia), np.array(g.ib), np.array(g.meas), np.array(g.info), np.array(g.kind))

def check(name, n_out, method):
    g = P.ReadG2O(os.path.join(ROOT, "tests/golden/data", name + ".g2o"))
    if n_out: g.add_random_C(n_out, 1)
    o = og(g)
    s = P.Solver(g, P.Options(method=method))
    c, r, J = s.evaluate()
    oc, orr, oJ = O.evaluate(o, method=method)
    print(f"[{name}+{n_out} m{method}] cost {c:.12e} oracle {oc:.12e} |dr| {np.abs(r-orr).max():.2e} |dJ| {np.abs(J-oJ).max():.2e}")
    gd, hd = s.normal_eq()
    ogd, ohd, _ = O.normal_eq(o, method=method)
    print(f"   |dg| {np.abs(gd-ogd).max():.2e} (|g| {np.abs(ogd).max():.2e})  |dHd| {np.abs(hd-ohd).max():.2e} (|Hd| {np.abs(ohd).max():.2e})")
    x = np.random.default_rng(0).standard_normal(3*g.n_poses)
    y = s.spmv(x)
    _, _, oy = O.normal_eq(o, method=method, x=x)
    print(f"   spmv |dy| {np.abs(y-oy).max():.2e} (|y| {np.abs(oy).max():.2e})")
    t = time.time(); summ = s.solve(); dt = time.time()-t
    d = summ.as_dict()
    print("   solve:", d["termination_name"], d["iterations"], "cost", d["final_cost"], "pcg", d["total_pcg_iters"], f"{dt:.2f}s",
          {k: round(v,3) for k,v in d.items() if k.startswith("seconds")})
    t = time.time(); ores = O.lm_direct(o, O.Options(method=method)); odt = time.time()-t
    xs = s.poses()
    print(f"   oracle direct: {O.TERM[ores.termination]} {ores.iterations} cost {ores.final_cost} {odt:.2f}s ; max|dxy| {np.abs(xs[:,:2]-ores.poses[:,:2]).max():.3e} max|dth| {np.abs(xs[:,2]-ores.poses[:,2]).max():.3e}")
    recs = s.iter_records()
    for a, b in list(zip(recs, ores.records))[:3] + list(zip(recs, ores.records))[-2:]:
        print("     ", a["iter"], a["step_ok"], f"{a['cost']:.10e} {b['cost']:.10e} rad {a['radius']:.3e} {b['radius']:.3e} pcg {a['pcg_iters']} rel {a['pcg_rel_residual']:.1e}")
    s.close()

if __name__ == "__main__":
    print(P.lib().pgo_version())
    check("INTEL", 50, 1)
    check("INTEL", 0, 0)
    check("MIT", 0, 1)
    if len(sys.argv) > 1:
        check("M3500", 0, 1)
