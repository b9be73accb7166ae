"""the CNN + LSTM case of scripts/bench_pixel.py alone (for rocprofv3)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_pixel
bench_pixel.run(256, 32, rnn=True, steps=2)
