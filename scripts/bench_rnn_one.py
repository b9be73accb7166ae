"""one shape of scripts/bench_rnn.py (for rocprofv3): python scripts/bench_rnn_one.py [B N D A]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_rnn
a = [int(v) for v in sys.argv[1:5]] or [64, 128, 17, 6]
bench_rnn.run(*a, steps=5)
