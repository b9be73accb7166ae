"""the host-fed learner by itself (bench.py's `host-fed learner` secondary): batches from host memory through the pinned
double-buffered ingest, from arrays, from per-step Python objects in one process, and from worker processes"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    import bench
    r = bench.secondary_host_fed()
    for k, v in r.items():
        print(k, '->', json.dumps(v) if isinstance(v, dict) else v)
