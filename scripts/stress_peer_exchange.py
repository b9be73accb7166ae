"""Repeat the peer-exchange protocol test (tests/test_gpu_dist.py: several processes sharing the one GPU) to shake out
ordering bugs that only show under contention:   python scripts/stress_peer_exchange.py 4 6   (world, repeats)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

if __name__ == '__main__':
    import test_gpu_dist as T
    world, reps = int(sys.argv[1]), int(sys.argv[2])
    fails = 0
    for i in range(reps):
        try:
            res = T._run_xchg(world, 'protocol')
            print('run %d ok: %.1f us per graph of 3 exchanges' % (i, res[0]['us_per_graph_of_3_exchanges']), flush=True)
        except AssertionError as e:
            fails += 1
            print('run %d FAILED: %s' % (i, str(e)[-600:]), flush=True)
    print('world %d: %d / %d failed' % (world, fails, reps))
    sys.exit(1 if fails else 0)
