"""Worker of tests/test_gpu_sharded.py: one rank of a sharded solve (run as a subprocess)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import toy_robust_backend_slam_amd as P  # noqa: E402


def main():
    cfg = json.loads(sys.argv[1])
    rank, world = cfg["rank"], cfg["world"]
    if cfg["graph"] == "synth":
        g = P.synth_manhattan(cfg["n_poses"], 4.0, 0.10, cfg["seed"])
    elif cfg["graph"] == "recipe":      # an arbitrary small graph of tests/test_gpu_fuzz.py
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_fuzz import make_graph
        g = P.Graph.from_arrays(*make_graph(*cfg["recipe"]))
    else:
        g = P.ReadG2O(os.path.join(ROOT, "tests", "golden", "data", cfg["graph"] + ".g2o"))
        if cfg.get("outliers"):
            g.add_random_C(cfg["outliers"], 1)
    if cfg.get("comm") == "rccl":  # world == 1 only on a one-GPU box (RCCL refuses duplicate devices)
        comm = P.Comm.rccl(P.Comm.unique_id(), rank, world, 0)
    else:
        comm = P.Comm.shm(cfg["name"], rank, world, 0) if world > 1 else None
    for k, v in (cfg.get("knobs") or {}).items():   # test hooks (pgo_debug_set_knob), read when the handle is created
        P.set_knob(k, v)
    s = P.Solver(g, P.Options(**cfg["options"]), comm, device=0)
    c0, _, _ = s.evaluate(want_r=False, want_J=False)
    summ = s.solve()
    out = dict(cost0=c0, summary=summ.as_dict(), records=s.iter_records(), info=s.info().as_dict())
    np.save(os.path.join(cfg["out"], "poses_%d.npy" % rank), s.poses())
    if cfg.get("chi2"):
        np.save(os.path.join(cfg["out"], "chi2_%d.npy" % rank), s.edge_chi2())
    json.dump(out, open(os.path.join(cfg["out"], "out_%d.json" % rank), "w"))
    s.close()
    if comm:
        comm.close()


if __name__ == "__main__":
    main()
