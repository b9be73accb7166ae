"""configs[1] with the MLP policy (64 x 128 x 17, A = 6): a few graph-replayed learns for a rocprofv3 kernel table"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
r = bench.secondary_ppo(64, 128, 17, 6, False, steps=20, cpu=False)
print({k: v for k, v in r.items() if not isinstance(v, dict)})
