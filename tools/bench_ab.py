"""A/B of the whole training step on ONE box: `python tools/bench_ab.py path/to/other/libstgcn_hip.so [bench.py flags]`
runs bench.py against another build of the library (boxes of the pool differ by a few per cent, so two builds are
only comparable inside one call)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
from stgcn_rt import lib as L  # noqa: E402

L._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
