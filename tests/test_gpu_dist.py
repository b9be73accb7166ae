"""GPU tier (-m gpu), world_size 2: the data-parallel path of PPOLearner on the HIP kernels.

A gpurun box has ONE GPU and RCCL refuses two ranks on one device, so the two ranks share cuda:0
and exchange through gloo (which stages device tensors through the host).  What this covers is
everything in the N > 1 path except the transport: the real kernels fed with all-reduced loss
partial sums / gradients / moments, `n_total = rows * world`, the rank-0-only log_var share of the
merged gradient all-reduce, the gathered value moments.  Each rank owns half of the
sub-trajectories and must reproduce the single reference learner's golden trace."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import helpers as H

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q):
    try:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        g, case = H.load_golden(name)
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
        assert learner.world_size == world and learner.use_graph        # graph segments between the collectives
        assert str(learner.device).startswith('cuda')
        stats = learner.learn(shard(batch))
        cpu = lambda t: t.detach().cpu().numpy().copy()  # noqa: E731
        out = {'stats': dict(stats), 'trace': learner.trace, 'adv': cpu(learner._ws.adv), 'ret': cpu(learner._ws.ret),
               'actor': cpu(learner.model.actor_flat), 'critic': cpu(learner.model.critic_flat),
               'z': {k: cpu(v) for k, v in learner.model.z_filter.state_dict().items()}
               if zstate is not None else None,
               'exp_counter': learner.exp_counter, 'segments': len(next(iter(learner._graphs.values())).items), 'collectives': learner.collectives_per_step,
               'fused': bool(getattr(learner._ws, 'fused', False))}
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # surface the failure in the parent
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))


# (cfg5_*: a 512-row shard of the benchmark shape per rank -- the fused row-block epoch kernels with the
# data-parallel right-hand sides, smx_ppo_epoch_combine_f32 after the all-reduce, the deferred tail exchange)
@pytest.mark.parametrize('name', ['tiny_clip', 'cfg2_adapt', 'cfg1_rnn_adapt', 'tiny_pixel_rnn_adapt', 'cfg5_clip',
                                  'cfg5_adapt', 'cfg5_adapt_earlyexit'])
def test_two_rank_hip_learner_equals_single_learner(name):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=240)
            res[r] = out
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    for r in range(world):
        assert 'error' not in res[r], res[r].get('error')
    g, case = H.load_golden(name)
    adv = np.concatenate([res[r]['adv'].reshape(-1) for r in range(world)]).reshape(g['advantages'].shape)
    ret = np.concatenate([res[r]['ret'].reshape(-1) for r in range(world)]).reshape(g['returns'].shape)
    np.testing.assert_allclose(adv, g['advantages'], atol=H.ATOL, rtol=H.RTOL)   # GLOBAL normalisation
    np.testing.assert_allclose(ret, g['returns'], atol=H.ATOL, rtol=H.RTOL)
    for r in range(world):
        H.assert_trace_close(res[r]['trace'], g, what='%s rank %d' % (name, r))
        H.assert_stats_close(res[r]['stats'], g, what='%s rank %d' % (name, r))
        assert res[r]['exp_counter'] == case['shape']['B']
    assert res[0]['segments'] >= 3
    if name.startswith('cfg5'):
        # one all-reduce per paired epoch + the advantage moments + the end-of-learn gather = 12 per learn;
        # this first learn also holds the workspace's batch-size exchange and the eager warm-up pass that
        # precedes the capture: 1 + 12 + 12
        per = case['hyper'].get('epoch_policy', 10) + 2
        assert res[0]['fused'] and res[0]['collectives'] == 1 + 2 * per, res[0]['collectives']
    # replicas stay bit-identical: same all-reduced gradients -> same Adam step everywhere
    np.testing.assert_array_equal(res[0]['actor'], res[1]['actor'])
    np.testing.assert_array_equal(res[0]['critic'], res[1]['critic'])
    if res[0]['z'] is not None:
        for k in ('running_sum', 'running_sumsq', 'count'):
            np.testing.assert_array_equal(res[0]['z'][k], res[1]['z'][k])
            np.testing.assert_allclose(res[0]['z'][k], g['zfinal.' + k], rtol=2e-6, atol=1e-3)


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N = 2 code path (launch contract, rank sharding, barrier + max-over-ranks timing,
    the single JSON line) rehearsed over gloo on one GPU; the number itself means nothing here"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMX_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1']
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['scaling'] == 'weak'
    assert out['config']['parallelism'] == 'dp2' and out['value'] > 0
    assert out['strong']['global_batch'] == 1024 and out['strong']['B_per_gpu'] == 512 and out['strong']['value'] > 0
    assert out['config']['collectives_per_step'] == 12 and out['config']['epoch_kernels'] == 'fused row-block'
    assert out['config']['epoch_all_reduce_us'] > 0 and out['config']['epoch_all_reduce_bytes'] > 1e6
    assert 'cpu_baseline' not in out and out['roofline']['frac'] > 0
    assert all(np.isfinite(v) for v in out['final_stats'].values())
