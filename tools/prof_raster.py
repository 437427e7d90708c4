"""Tiny driver for rocprofv3: a few raster forward/backward launches at one shape."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import bench
N, sub, IS, TS = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 3, 512, 1)
print(bench(N, sub, IS, TS, iters=3))
