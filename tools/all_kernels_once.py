#!/usr/bin/env python
"""Launch every kernel family a few times (for one rocprofv3 --kernel-trace --stats summary, profiles/r1_all_kernels_stats.txt)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench, ice_bench, turb_bench, init_bench  # noqa: E402

sys.argv = ["kbench", "--iters", "5", "--reps", "3"]
kbench.main()
sys.argv = ["kbench", "--iters", "5", "--reps", "3", "--precision", "f32", "--algos", "ecmwf,coare3p6"]
kbench.main()
ice_bench.main()
turb_bench.main()
init_bench.main()
