import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--aux", ""]
bench.main()
