"""dev helper: the bench's short form against an alternative build of the library.  python tools/r3_altlib.py <lib.so> [bench args]"""
import sys, os, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ngspeciesid_amd import runtime
runtime.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
