"""Build the experiment library next to the product one: toy-robust-backend-slam_amd/libpgo_exp.so = the same sources with
-DPGO_EXPERIMENTS (environment switches PGO_*; never the default: the exp_*.sh scripts select it with PGO_LIB).  Run here, before
the gpurun call: the .so travels with the snapshot.  usage: build_exp.py [EXTRA_DEFINE ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
b = importlib.import_module("toy-robust-backend-slam_amd._build")
print(b.build_lib(force=True, defines=("PGO_EXPERIMENTS",) + tuple(sys.argv[1:]), out=b.LIB.replace("libpgo.so", "libpgo_exp.so")))
