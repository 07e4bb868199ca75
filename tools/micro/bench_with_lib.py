"""dev helper: run bench.py against another build of the library (timing builds of tools/micro/build_repeat.sh).
    python tools/micro/bench_with_lib.py build_alt/libngsid_hip_rep2.so --steps 1 --warmup 1 --no-cpu-baseline --no-cli"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ngspeciesid_amd import runtime
runtime.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
