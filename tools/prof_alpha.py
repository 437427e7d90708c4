import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv.append("--alpha-import")
from tools.microbench import bench_alpha
print(bench_alpha(64, 3, 512, iters=3))
