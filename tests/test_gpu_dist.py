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
import sys

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


def _worker(rank, world, port, name, q, cuts=None, overrides=None):
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
        if cuts is not None:
            lo, hi = cuts[rank], cuts[rank + 1]

        def shard(x):
            if isinstance(x, dict):
                return type(x)((k, shard(v)) for k, v in x.items())
            if isinstance(x, list):
                return [shard(v) for v in x]
            return x[lo:hi] if x is not None else None
        case_local = copy.deepcopy(case)
        case_local['shape']['B'] = hi - lo
        learner = H.make_learner(case_local, params, zstate, session_overrides=overrides)
        assert learner.world_size == world and learner.use_graph        # one graph, or segments between the collectives
        assert str(learner.device).startswith('cuda')
        stats = learner.learn(shard(batch))
        cpu = lambda t: t.detach().cpu().numpy().copy()  # noqa: E731
        out = {'stats': dict(stats), 'trace': learner.trace, 'adv': cpu(learner._ws.adv), 'ret': cpu(learner._ws.ret),
               'actor': cpu(learner.model.actor_flat), 'critic': cpu(learner.model.critic_flat),
               'params': {k: v.copy() for k, v in learner.model.numpy_params().items()} if rank == 0 else None,
               'z': {k: cpu(v) for k, v in learner.model.z_filter.state_dict().items()}
               if zstate is not None else None,
               'exp_counter': learner.exp_counter, 'segments': len(next(iter(learner._graphs.values())).items), 'collectives': learner.collectives_per_step,
               'fused': bool(getattr(learner._ws, 'fused', False)),
               'exchange': getattr(learner, 'exchange_kind', 'process group')}
        q.put((rank, out))
        dist.barrier()
        if learner._dist.exchange is not None:
            learner._dist.exchange.close()
        dist.destroy_process_group()
    except Exception:  # surface the failure in the parent
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))


def _run_ranks(name, world, cuts=None, overrides=None):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, cuts, overrides)) for r in range(world)]
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
    adv = np.concatenate([res[r]['adv'].reshape(-1) for r in range(world)])
    ret = np.concatenate([res[r]['ret'].reshape(-1) for r in range(world)])
    H.assert_adv_ret(adv, ret, g, case['shape']['B'])                             # GLOBAL normalisation
    for r in range(world):
        H.assert_trace_close(res[r]['trace'], g, what='%s rank %d' % (name, r))
        H.assert_stats_close(res[r]['stats'], g, what='%s rank %d' % (name, r))
        assert res[r]['exp_counter'] == case['shape']['B']
    # replicas stay bit-identical: same all-reduced gradients -> same Adam step everywhere
    for r in range(1, world):
        np.testing.assert_array_equal(res[0]['actor'], res[r]['actor'])
        np.testing.assert_array_equal(res[0]['critic'], res[r]['critic'])
        if res[0]['z'] is not None:
            for k in ('running_sum', 'running_sumsq', 'count'):
                np.testing.assert_array_equal(res[0]['z'][k], res[r]['z'][k])
    if res[0]['z'] is not None:
        for k in ('running_sum', 'running_sumsq', 'count'):
            np.testing.assert_allclose(res[0]['z'][k], g['zfinal.' + k], rtol=2e-6, atol=1e-3)
    # the replicas' updated parameters against the single learner's of the reference, every element
    H.assert_final_params_dict(res[0]['params'], g, what='%s %d ranks' % (name, world))
    return res, case


# (cfg5_*: a 512-row shard of the benchmark shape per rank -- the fused row-block epoch kernels with the
# data-parallel right-hand sides, smx_ppo_epoch_combine_f32 after the all-reduce, the deferred tail exchange)
@pytest.mark.parametrize('name', ['tiny_clip', 'cfg2_adapt', 'cfg1_rnn_adapt', 'tiny_pixel_rnn_adapt', 'cfg5_clip',
                                  'cfg5_adapt', 'cfg5_adapt_earlyexit', 'cfg2_rnn_adapt', 'cfg2_rnn_clip', 'cfg5_rnn_adapt',
                                  'cfg5_rnn_clip', 'b1024_d17_rnn_adapt'])
def test_two_rank_hip_learner_equals_single_learner(name):
    """the default at N > 1: the fp32 exchanges run as kernels over IPC-mapped peer buffers (PeerExchange,
    set up and self-checked against the process group when the workspace is built), so the whole learn is ONE
    captured graph again"""
    res, case = _run_ranks(name, 2)
    assert res[0]['exchange'].startswith('peer buffers'), res[0]['exchange']
    if name.startswith('cfg5') and 'rnn' not in name:
        # every exchange of the learn is a kernel inside the one graph; the count per learn is unchanged:
        # one all-reduce per paired epoch + the advantage moments + the end-of-learn gather = 12 (this first
        # learn also holds the workspace's batch-size exchange and the eager warm-up pass: 1 + 12 + 12)
        per = case['hyper'].get('epoch_policy', 10) + 2
        assert res[0]['fused'] and res[0]['segments'] == 1, res[0]['segments']
        assert res[0]['collectives'] >= 2 * per, res[0]['collectives']


@pytest.mark.parametrize('name', ['cfg2_adapt', 'cfg5_adapt', 'cfg1_rnn_adapt', 'cfg2_rnn_clip', 'tiny_pixel_rnn_adapt'])
def test_two_rank_hip_learner_on_the_process_group(name):
    """session_config.learner.peer_exchange = False: the collectives stay on torch.distributed (RCCL on a real node,
    gloo here) between hipGraph segments -- what a failed self-check falls back to"""
    res, case = _run_ranks(name, 2, overrides={'peer_exchange': False})
    assert res[0]['exchange'] == 'process group' and res[0]['segments'] >= 3
    if name.startswith('cfg5'):
        per = case['hyper'].get('epoch_policy', 10) + 2
        assert res[0]['fused'] and res[0]['collectives'] == 1 + 2 * per, res[0]['collectives']


@pytest.mark.parametrize('name', ['ragged_adapt_offpolicy', 'ragged_clip'])
def test_three_ranks_with_different_block_counts_hip(name):
    """37 sub-trajectories cut 5 | 20 | 12: 1, 2 and 1 sixteen-row loss blocks per rank (the loss-partial rows a
    rank does not own are cleared before every exchange)"""
    _run_ranks(name, 3, cuts=[0, 5, 25, 37])


def _ddpg_worker(rank, world, port, name, q, overrides):
    try:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
        from surreal_amd import synthetic
        import ddpg_helpers as DH
        g, case = DH.load(name)
        B = case['B']
        lo, hi = rank * B // world, (rank + 1) * B // world
        local = copy.deepcopy(case)
        local['B'] = hi - lo
        L = DH.make_learner(local)
        for k, v in (overrides or {}).items():
            L.session_config.learner[k] = v
        assert L.world_size == world and str(L.device).startswith('cuda')
        trace = []
        for it in range(case['iters']):
            b = synthetic.make_ddpg_batch(B, case['D'], case['A'], seed=10 + it)

            def cut(x):
                if isinstance(x, dict):
                    return {k: cut(v) for k, v in x.items()}
                return x[lo:hi]
            trace.append(dict(L.learn(cut(b))))
        q.put((rank, {'trace': trace, 'params': L.model.numpy_params(), 'target': L.model_target.numpy_params(),
                      'exchange': L.exchange_kind, 'graph': L._ws.graph is not None}))
        dist.barrier()
        if L._dist.exchange is not None:
            L._dist.exchange.close()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))


@pytest.mark.parametrize('name,peer', [('cfg3_cheetah512', True), ('tiny_hard', True), ('cfg3_cheetah512', False)])
def test_two_rank_hip_ddpg_equals_single_learner(name, peer):
    """data-parallel DDPG on the HIP kernels (surreal/learner/ddpg.py:244-400 run on two shards of every batch): the
    gradients and the reported means are all-reduced before each Adam step -- as kernels over the peer buffers INSIDE
    the one captured graph of the iteration (peer=True), or eagerly on the process group (peer=False) -- and two ranks
    reproduce the single reference learner's trace with bit-identical replicas"""
    import json
    import ddpg_helpers as DH
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddpg_worker, args=(r, world, port, name, q, None if peer else {'peer_exchange': False}))
             for r in range(world)]
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
    g, case = DH.load(name)
    ref = json.loads(str(g['trace_json']))
    for r in range(world):
        if peer:
            assert res[r]['exchange'].startswith('peer buffers') and res[r]['graph'], (res[r]['exchange'], res[r]['graph'])
        else:
            assert res[r]['exchange'] == 'process group' and not res[r]['graph']
        for it, want in enumerate(ref):
            for k, v in want.items():
                np.testing.assert_allclose(res[r]['trace'][it][k], v, atol=1e-5, rtol=1e-5,
                                           err_msg='%s rank %d iteration %d %s' % (name, r, it, k))
    for k in res[0]['params']:
        np.testing.assert_array_equal(res[0]['params'][k], res[1]['params'][k])
        np.testing.assert_array_equal(res[0]['target'][k], res[1]['target'][k])
    DH._assert_params(name + ' 2 ranks', 'model', res[0]['params'], g, 'final.')        # every element (1e-5)
    DH._assert_params(name + ' 2 ranks', 'target', res[0]['target'], g, 'target.')


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
    assert out['config']['exchange'].startswith('peer buffers') and not out['config']['graph_segments']
    assert 0.0 < out['exchange_model']['predicted_efficiency'] <= 1.0
    assert 'cpu_baseline' not in out and out['roofline']['frac'] > 0
    assert all(np.isfinite(v) for v in out['final_stats'].values())


def test_bench_launches_its_own_ranks_and_always_prints_a_line():
    """`python bench.py --gpus 2` with no launcher around it: over gloo it starts its own two ranks and prints the
    measurement line; over RCCL on a box with fewer devices than ranks it prints ONE diagnostic line (value null,
    the reason) and exits cleanly -- first contact with a node never ends in an assert or a hang"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1']
    r = subprocess.run(cmd, env=dict(env, SMX_BENCH_BACKEND='gloo'), cwd=root, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['exchange'].startswith('peer buffers')
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=120)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
        out = json.loads(lines[0])
        assert out['value'] is None and 'one device per rank' in out['error'] and out['n_gpus'] == 2


# ---- PeerExchange by itself: two processes share the one GPU (the protocol, the bounded spins and graph capture are
# exercised; NOT the cross-device cache behaviour -- both ranks sit behind one L2 -- which is why PeerExchange.create
# self-checks against the process group on the real node before the learner uses it) --------------------------------
def _xchg_worker(rank, world, port, q, mode):
    try:
        import time
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from surreal_amd.distributed.peer_exchange import PeerExchange
        n = 540001                                   # the benchmark's per-epoch exchange is 536 k floats (2.1 MB)
        out = {}
        if mode == 'protocol':
            ex = PeerExchange.create(dist, n, timeout_s=20.0, rounds=12)
            assert ex is not None, 'peer exchange could not be set up on one GPU'
            out['kind'] = ex.memory_kind
            # fixed-order host sum: rank 0's vector + rank 1's vector + ..., in fp32, is what every rank must hold
            gens = [torch.Generator().manual_seed(77 + r) for r in range(world)]
            err = torch.zeros(1, dtype=torch.int32, device='cuda')
            for it, m in enumerate((n, 1, 5, 4096, 131071)):
                parts = [torch.randn(m, generator=g) for g in gens]
                want = parts[0].clone()
                for p in parts[1:]:
                    want += p
                t = parts[rank].cuda()
                ex.all_reduce(t, err=err)
                assert torch.equal(t.cpu(), want), 'all-reduce of %d floats is not the rank-ordered fp32 sum' % m
                got = torch.empty(world * min(m, n // world), device='cuda')
                ex.all_gather_into_tensor(got, parts[rank][:min(m, n // world)].cuda().contiguous(), err=err)
                assert torch.equal(got.cpu(), torch.cat([p[:min(m, n // world)] for p in parts]))
            # captured in a hipGraph and replayed: the sequence number lives in the buffer
            a = torch.zeros(n, device='cuda')
            b = torch.zeros(3 * world, device='cuda')
            src = torch.full((n,), float(rank + 1), device='cuda')
            src3 = torch.full((3,), float(rank + 1), device='cuda')
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                def step():
                    a.copy_(src)
                    ex.all_reduce(a, err=err)
                    ex.all_gather_into_tensor(b, src3, err=err)
                    a.mul_(0.5)
                    ex.all_reduce(a, err=err)
                step()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                    step()
                for it in range(24):
                    # new values every replay: a peer that read the staging before the owner's stores landed would
                    # see the previous exchange's numbers
                    src.fill_(float(rank + 1 + it))
                    src3.fill_(float(rank + 1 + it))
                    g.replay()
                    torch.cuda.synchronize()
                    tot = sum(r + 1 + it for r in range(world))
                    bad = (a != 0.5 * tot * world).nonzero().flatten()
                    assert bad.numel() == 0, 'replay %d: %d of %d elements wrong, first at %d: %r (want %r), err %#x' % (
                        it, bad.numel(), n, int(bad[0]), float(a[bad[0]]), 0.5 * tot * world, int(err.item()))
                    assert b.cpu().tolist() == [float(r + 1 + it) for r in range(world) for _ in range(3)]
                # how long one 2.1 MB all-reduce takes when the peers arrive together (one GPU: protocol cost only)
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(50):
                    g.replay()
                torch.cuda.synchronize()
                out['us_per_graph_of_3_exchanges'] = (time.perf_counter() - t0) / 50 * 1e6
            done, e = ex.status()
            assert e == 0 and int(err.item()) == 0
            out['exchanges'] = done
            dist.barrier()
            ex.close()
        elif mode == 'setup_failure':
            # one rank cannot allocate its buffer: EVERY rank must come back from create() with None (the process-group
            # fallback), none may be left inside a collective
            from surreal_amd import _lib as LL
            if rank == 1:
                real = LL.call

                def failing(name, *a):
                    if name == 'smx_xchg_alloc':
                        raise LL.SmxError('injected allocation failure')
                    return real(name, *a)
                LL.call = failing
                import surreal_amd.distributed.peer_exchange as PX
                PX.L.call = failing
            t0 = time.perf_counter()
            ex = PeerExchange.create(dist, 4096, timeout_s=1.0, rounds=4)
            out['none'] = ex is None
            out['s'] = time.perf_counter() - t0
            dist.barrier()
        else:
            # a peer that never shows up: the wait is bounded, the error words say who was missing, later waits
            # return at once, nothing hangs
            ex = PeerExchange(dist, 4096, timeout_s=0.3)
            err = torch.zeros(1, dtype=torch.int32, device='cuda')
            t = torch.ones(4096, device='cuda')
            if rank == 0:
                t0 = time.perf_counter()
                ex.all_reduce(t, err=err)
                torch.cuda.synchronize()
                first = time.perf_counter() - t0
                t0 = time.perf_counter()
                ex.all_reduce(t, err=err)
                torch.cuda.synchronize()
                out['first_s'], out['second_s'] = first, time.perf_counter() - t0
                out['err'] = int(err.item())
                out['status'] = ex.status()
            dist.barrier()
            ex.close()
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))


def _run_xchg(world, mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, world, port, q, mode)) for r in range(world)]
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
    return res


@pytest.mark.parametrize('world', [2, 4, 8])
def test_peer_exchange_is_the_rank_ordered_sum_and_replays_in_a_graph(world):
    res = _run_xchg(world, 'protocol')
    print('\\npeer exchange, %d ranks on one GPU: memory %s, %.1f us per captured graph of 3 exchanges (2.1 MB all-reduce x 2 '
          '+ a 12-byte all-gather)' % (world, res[0]['kind'], res[0]['us_per_graph_of_3_exchanges']))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(d):
        import json
        json.dump(res[0], open(os.path.join(d, 'peer_exchange_w%d.json' % world), 'w'))


def test_peer_exchange_setup_failure_on_one_rank_falls_back_on_all():
    res = _run_xchg(2, 'setup_failure')
    assert res[0]['none'] and res[1]['none'], res
    assert res[0]['s'] < 20 and res[1]['s'] < 20, res


def test_peer_exchange_wait_is_bounded_and_reports_the_missing_peer():
    res = _run_xchg(2, 'absent')
    r0 = res[0]
    assert 0.25 < r0['first_s'] < 5.0, r0          # the 0.3 s bound, not a hang
    assert r0['second_s'] < 0.25, r0                # once the error is up nobody waits again
    assert r0['err'] & 0x100 and (r0['err'] & 0xf) == 1, hex(r0['err'])        # timeout, peer 1
    assert r0['status'][1] == r0['err']
