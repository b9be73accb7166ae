#!/usr/bin/env python
"""First contact with an N-GPU node in under a minute: everything the data-parallel learner needs from the node, checked
one step at a time, each with its own verdict -- so that a failure names the layer it belongs to instead of showing up
as a hung benchmark.

    python scripts/selftest_multigpu.py --gpus 8          (launches its own ranks; or under torch.distributed.run)
    SMX_BENCH_BACKEND=gloo python scripts/selftest_multigpu.py --gpus 2     (rehearsal: the ranks share one GPU)

Steps (every rank takes part in every step whatever it has seen: the process-group calls stay in step):
  1. the process group (RCCL over xGMI; gloo in the rehearsal): init + one all-reduce, bounded by a timeout
  2. device-pair IPC: every rank allocates an exchange buffer (smx_xchg_alloc), exports it, opens every peer's handle
  3. the exchange self-check: 16 rounds of all-reduce / all-gather through the peer buffers against the process
     group's results, and bit-equality of the result across ranks (what must hold for replicas to stay identical:
     surreal/learner/ppo.py:413-416, 541-562 run on shards)
  4. one all-reduce of the epoch payload (536 k floats), microseconds through both paths
  5. one data-parallel learn() of the benchmark shape split over the ranks against the single-learner golden
     (tests/golden/ppo_cfg5_adapt.npz): advantages / losses / KL at 1e-5
Writes profiles/multigpu_selftest.json (rank 0) and prints it."""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def launch(args):
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    try:
        return subprocess.run(cmd, timeout=args.budget, cwd=ROOT).returncode
    except subprocess.TimeoutExpired:
        print(json.dumps({'ok': False, 'error': 'the %d-rank self-test exceeded %d s' % (args.gpus, args.budget)}))
        return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=2)
    ap.add_argument('--budget', type=int, default=240, help='wall-clock seconds for the whole self-test')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'multigpu_selftest.json'))
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ:
        return launch(args)
    import numpy as np
    import torch
    import torch.distributed as dist
    world, rank, local = int(os.environ['WORLD_SIZE']), int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('SMX_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    rep = {'world': world, 'backend': backend, 'devices_visible': ndev, 'steps': {}, 'ok': False,
           'rehearsal_on_shared_gpu': backend != 'nccl'}
    t_all = time.time()

    def finish(code=0):
        rep['seconds'] = time.time() - t_all
        if rank == 0:
            os.makedirs(os.path.dirname(args.out), exist_ok=True)
            json.dump(rep, open(args.out, 'w'), indent=1)
            print(json.dumps(rep), flush=True)
        return code

    if ndev == 0 or (backend == 'nccl' and world > ndev):
        rep['error'] = '%d visible GPU(s) for %d ranks (RCCL needs one device per rank)' % (ndev, world)
        return finish(0)
    torch.cuda.set_device(local % ndev)
    dev = torch.device('cuda', local % ndev)
    # ---- 1. process group ------------------------------------------------------------------------------
    t0 = time.time()
    try:
        tmo = datetime.timedelta(seconds=60)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        one = torch.ones(4, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(one)
        ok = bool((one == world).all())
        rep['steps']['1 process group'] = {'ok': ok, 'seconds': time.time() - t0}
    except Exception as e:
        rep['steps']['1 process group'] = {'ok': False, 'error': repr(e)}
        return finish(1)
    from surreal_amd.distributed.peer_exchange import PeerExchange
    n = 540000
    # ---- 2. + 3. IPC set-up and the self-check (collective; failures are agreed on by all ranks) -----------
    t0 = time.time()
    ex = PeerExchange.create(dist, n, timeout_s=5.0, rounds=16)
    rep['steps']['2 ipc + 3 self-check'] = {'ok': ex is not None, 'seconds': time.time() - t0,
                                            'message': ex.check_message if ex is not None else PeerExchange.last_failure,
                                            'memory': ex.memory_kind if ex is not None else None}
    # ---- 4. the epoch exchange, microseconds through both paths -------------------------------------------
    v = torch.randn(n, device=dev)

    def time_it(fn, reps=20):
        for _ in range(3):
            fn(v)
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(v)
        e1.record(); torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / reps * 1e3], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)
    try:
        step4 = {'floats': n, 'process_group_us': time_it(dist.all_reduce)}
        if ex is not None:
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            step4['peer_exchange_us'] = time_it(lambda t: ex.all_reduce(t, err=err))
            step4['peer_exchange_error_word'] = int(err.item())
        step4['ok'] = True
    except Exception as e:
        step4 = {'ok': False, 'error': repr(e)}
    rep['steps']['4 epoch all-reduce'] = step4
    if ex is not None:
        dist.barrier()
        ex.close()
    # ---- 5. one sharded learn of the benchmark shape against the single-learner golden ---------------------
    t0 = time.time()
    try:
        import copy
        import helpers as H
        g, case = H.load_golden('cfg5_adapt')
        batch, params, zstate = H.case_inputs(case)
        B = case['shape']['B']
        lo, hi = rank * B // world, (rank + 1) * B // world

        def shard(x):
            if isinstance(x, dict):
                return type(x)((k, shard(v)) for k, v in x.items())
            if isinstance(x, list):
                return [shard(v) for v in x]
            return x[lo:hi] if x is not None else None
        case_local = copy.deepcopy(case)
        case_local['shape']['B'] = hi - lo
        learner = H.make_learner(case_local, params, zstate)
        stats = learner.learn(shard(batch))
        H.assert_trace_close(learner.trace, g, what='cfg5_adapt over %d ranks' % world)
        ok5, msg5 = True, getattr(learner, 'exchange_kind', None)
        kl = float(stats['_pol_kl'])
    except Exception as e:
        ok5, msg5, kl = False, repr(e)[:500], None
    flag = torch.tensor([1.0 if ok5 else 0.0], device=dev if backend == 'nccl' else 'cpu')
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    rep['steps']['5 sharded learn vs golden'] = {'ok': bool(flag.item() >= 1.0), 'rank0': msg5, 'pol_kl': kl,
                                                 'seconds': time.time() - t0}
    rep['ok'] = all(s.get('ok') for s in rep['steps'].values())
    dist.barrier()
    dist.destroy_process_group()
    return finish(0 if rep['ok'] else 1)


if __name__ == '__main__':
    sys.exit(main())
