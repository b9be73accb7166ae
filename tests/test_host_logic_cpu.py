"""CPU tier: the product PPOLearner's HOST logic (epoch control, device-side early-exit
protocol, statistics slots, forward reuse, Adam step bookkeeping) driven through the torch-CPU
kernel test double must reproduce the reference's golden traces.  This isolates orchestration
bugs from kernel bugs; the kernels themselves are checked on the GPU (-m gpu)."""
import copy

import numpy as np
import pytest
import torch

import helpers as H

ALL_CASES = H.golden_cases(big=False)        # MLP policy and LSTM-stem policy (the reference default)


@pytest.mark.parametrize('name', ALL_CASES)
def test_learner_host_logic_matches_reference(name, cpu_double):
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate)
    stats = learner.learn(copy.deepcopy(batch))
    ws = learner._ws
    np.testing.assert_allclose(ws.adv.numpy().reshape(g['advantages'].shape), g['advantages'],
                               atol=H.ATOL, rtol=H.RTOL)
    np.testing.assert_allclose(ws.ret.numpy().reshape(g['returns'].shape), g['returns'],
                               atol=H.ATOL, rtol=H.RTOL)
    H.assert_trace_close(learner.trace, g, what=name)
    H.assert_stats_close(stats, g, what=name)
    H.assert_final_params(learner, g, case, what=name)
    if zstate is not None:
        sd = learner.model.z_filter.state_dict()
        for k in ('running_sum', 'running_sumsq', 'count'):
            np.testing.assert_allclose(sd[k].numpy(), g['zfinal.' + k], rtol=1e-6)


def test_layered_schedule_host_logic(cpu_double):
    """fused_epochs = False: the one-launch-per-layer lock-step epochs (what shapes outside the row-block kernels take)"""
    g, case = H.load_golden('tiny_adapt_cutoff2')
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate, session_overrides={'fused_epochs': False})
    stats = learner.learn(copy.deepcopy(batch))
    H.assert_trace_close(learner.trace, g, what='layered')
    H.assert_stats_close(stats, g, what='layered')


def test_second_learn_continues_adam_state(cpu_double):
    """Adam moments and step counters persist across learn() calls, as torch.optim does"""
    import ppo_oracle
    g, case = H.load_golden('tiny_clip')
    batch, params, zstate = H.case_inputs(case)
    hyper = dict(case['hyper'])
    hyper['n_step'] = case['shape']['N']
    O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate,
                                    **hyper)
    learner = H.make_learner(case, params, zstate)
    for it in range(3):
        b = dict(batch)
        so = O.learn(copy.deepcopy(b))
        sl = learner.learn(copy.deepcopy(b))
        for k in so:
            if k != '_lr':
                np.testing.assert_allclose(sl[k], so[k], atol=H.ATOL, rtol=H.RTOL,
                                           err_msg='iter %d %s' % (it, k))


def test_post_publish_adapts_coefficients(cpu_double):
    """beta / clip-epsilon adaptation and the reference-policy refresh (ppo.py:637-666)"""
    g, case = H.load_golden('tiny_adapt')
    batch, params, zstate = H.case_inputs(case)
    L = H.make_learner(case, params, zstate)
    L.learn(copy.deepcopy(batch))
    L.kl_record = [L.kl_target * 3.0]
    L._post_publish()
    assert L.beta == pytest.approx(1.5) and L.kl_record == [] and L.exp_counter == 0
    L.kl_record = [L.kl_target * 0.1]
    L._post_publish()
    assert L.beta == pytest.approx(1.0)
    np.testing.assert_array_equal(L.ref_target_model.actor_flat.numpy(), L.model.actor_flat.numpy())
    case2 = copy.deepcopy(case)
    case2['hyper']['ppo_mode'] = 'clip'
    L = H.make_learner(case2, params, zstate)
    L.kl_record = [1.0]
    L._post_publish()
    assert L.clip_epsilon == pytest.approx(0.2 / 1.2)
    L.kl_record = [0.0]
    L._post_publish()
    L.kl_record = [0.0]
    L._post_publish()
    assert L.clip_epsilon == pytest.approx(0.2 * 1.2)


@pytest.mark.parametrize('name', ['tiny_pixel_clip', 'tiny_pixel_rnn_adapt'])
def test_cnn_stem_chunked_critic_pass_with_a_tail(name, cpu_double):
    """the critic pass runs the CNN stem in chunks of cnn_chunk_frames; a chunk size that does not
    divide B*(N+1) leaves a shorter tail chunk inside the same workspace"""
    g, case = H.load_golden(name)
    batch, params, zstate = H.case_inputs(case)
    frames = case['shape']['B'] * (case['shape']['N'] + 1)
    chunk = max(2, frames // 3 + 1)
    assert frames % chunk != 0
    learner = H.make_learner(case, params, zstate, session_overrides={'cnn_chunk_frames': chunk})
    stats = learner.learn(copy.deepcopy(batch))
    H.assert_trace_close(learner.trace, g, what=name)
    H.assert_stats_close(stats, g, what=name)


def test_deferred_stats_and_scalar_recorder():
    """DeferredStats resolves once, when first looked at; ScalarRecorder stores it untouched and folds
    it into `latest` on demand (also when old history is trimmed); it pickles as a plain dict"""
    import pickle
    from surreal_amd.learner.base import DeferredStats, ScalarRecorder
    calls = []

    def make(value):
        box = {}

        def resolve():
            calls.append(value['k'])
            box['d']._value = dict(value)
        box['d'] = DeferredStats(resolve)
        return box['d']
    rec = ScalarRecorder(keep=3)
    a, b = make({'k': 1, 'x': 1.0}), make({'k': 2, 'y': 2.0})
    rec.add_scalars(a, 1)
    rec.add_scalars({'k': 0, 'z': 9.0}, 2)
    rec.add_scalars(b, 3)
    assert calls == []                                   # nothing has been looked at
    assert 'x' in a and calls == [1] and a['x'] == 1.0 and len(a) == 2 and calls == [1]
    assert rec.latest == {'k': 2, 'x': 1.0, 'z': 9.0, 'y': 2.0} and calls == [1, 2]
    c = make({'k': 3, 'w': 4.0})
    rec.add_scalars(c, 4)
    rec.add_scalars({'k': 5}, 5)                         # trims: everything dropped was folded first
    assert len(rec.history) == 3 and rec.latest['w'] == 4.0 and rec.latest['k'] == 5 and rec.latest['x'] == 1.0
    assert pickle.loads(pickle.dumps(c)) == {'k': 3, 'w': 4.0} and c.copy() == dict(c)
    assert 'in flight' in repr(make({'k': 7}))


@pytest.mark.parametrize('sk', [True, False])
@pytest.mark.parametrize('u8,C,H,W,feat', [(True, 3, 20, 20, 8), (False, 2, 36, 28, 24), (True, 3, 21, 22, 12)])
def test_cnn_stem_host_logic_matches_aten_autograd(cpu_double, u8, C, H, W, feat, sk):
    """CnnStem.forward / backward on the kernel double against torch.nn's Conv2d-ReLU-Conv2d-ReLU-Flatten-Linear-ReLU and
    autograd: the path selection (implicit-GEMM entry points for uint8 frames inside their limits -- with and without
    the partial-sum workspace --, materialised patches for fp32 frames or a width that is not a multiple of 4), the
    channel-last flatten order of the Linear and the parameter / gradient layout"""
    import torch.nn as nn
    from surreal_amd import kernels as KN
    from surreal_amd.model.cnn_stem import CnnParams, CnnStem
    torch.manual_seed(H + C)
    F = 4
    ref = nn.Sequential(nn.Conv2d(C, 16, 8, 4), nn.ReLU(), nn.Conv2d(16, 32, 4, 2), nn.ReLU(), nn.Flatten())
    with torch.no_grad():
        n_flat = ref(torch.zeros(1, C, H, W)).shape[1]
    fc = nn.Linear(n_flat, feat)
    frames = torch.randint(0, 256, (F, C, H, W), dtype=torch.uint8)
    if not u8:
        frames = frames.float()
    y = torch.relu(fc(ref(frames.float() / 255.0)))
    dy = torch.randn(F, feat)
    (y * dy).sum().backward()
    flat = torch.zeros(CnnParams.count((C, H, W), feat))
    p = CnnParams(flat, 0, (C, H, W), feat)
    src = {'conv1.W': ref[0].weight, 'conv1.b': ref[0].bias, 'conv2.W': ref[2].weight,
           'conv2.b': ref[2].bias, 'fc.W': fc.weight, 'fc.b': fc.bias}
    for k, v in p.views.items():
        v.copy_(src[k].detach())
    K = KN.default_kernels()
    stem = CnnStem(K)
    ws = stem.workspace(p, F, 'cpu')
    if sk:
        ws.sk = torch.empty(max(1, K.conv_u8_wgrad_ws_floats(p.c1, p.K1), K.conv_cl_wgrad_ws_floats(p.c2, p.k2)))
    xin = torch.zeros(F, 3 + feat)
    stem.forward(p, frames, F, ws, xin[:, 3:])
    implicit = u8 and W % 4 == 0
    assert (ws.cols1 is None) == implicit          # no patch matrix on the implicit forward path
    np.testing.assert_allclose(xin[:, 3:].numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-5)
    dxin = torch.zeros_like(xin)
    dxin[:, 3:] = dy * (y.detach() > 0)
    grads = torch.zeros_like(flat)
    stem.backward(p, F, ws, dxin[:, 3:], grads)
    assert (ws.cols1 is None) == (implicit and sk)
    gp = CnnParams(grads, 0, (C, H, W), feat)
    for k, v in gp.views.items():
        scale = float(src[k].grad.abs().max())
        np.testing.assert_allclose(v.numpy() / scale, src[k].grad.numpy() / scale, rtol=1e-4, atol=1e-5, err_msg=k)


def test_bench_without_devices_prints_one_diagnostic_line():
    """bench.py --gpus 2 on a host that shows no GPU: one JSON line with value null and the reason, exit code 0"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['HIP_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, cwd=root, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1000:] + r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out['value'] is None and out['n_gpus'] == 2 and 'GPU' in out['error']


def test_philox_restatement_meets_the_random123_known_answer_vectors():
    """tests/philox_ref.py (the numpy statement of the replay sampler's generator) against Random123's published
    philox4x32-10 vectors; the GPU tier holds the device code to the same vectors and to this restatement"""
    import philox_ref as P
    for ctr, key, want in P.KAT:
        assert P.philox4x32_10(ctr, key) == want
    idx = P.uniform_indices(4096, 1000, 1234, 0)
    assert idx.min() >= 0 and idx.max() < 1000 and len(set(idx.tolist())) > 900


def test_rnn_hidden_not_a_multiple_of_four_is_padded_inside(cpu_double):
    """rnn_hidden = 10 (ppo_net.py:144-149 takes any size): the stem is padded to 12 inside the parameter layout; outside --
    parameter dict, state dict, the agents' cells -- nothing shows it, and the pad stays exactly zero through a learn"""
    import copy
    import torch
    g, case = H.load_golden('tiny_rnn_h10_adapt')
    batch, params, zstate = H.case_inputs(case)
    learner = H.make_learner(case, params, zstate)
    m = learner.model
    assert m.rnn_hidden == 12 and m.rnn_hidden_logical == 10
    got = m.numpy_params()
    for k, v in params.items():
        assert got[k].shape == np.asarray(v).shape, (k, got[k].shape)
        np.testing.assert_array_equal(got[k], v)
    sd = m.state_dict()
    assert tuple(sd['rnn.weight_hh'].shape) == (40, 10) and tuple(sd['actor.fc1.W'].shape) == (24, 10)
    twin = H.make_learner(case, params, zstate).model
    twin.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items()})
    assert torch.equal(twin.flat, m.flat)
    learner.learn(copy.deepcopy(batch))
    pad = m.rnn.views['weight_hh'].view(4, 12, 12)
    assert float(pad[:, 10:, :].abs().max()) == 0.0 and float(pad[:, :, 10:].abs().max()) == 0.0
    assert float(m.actor.views['W1'][:, 10:].abs().max()) == 0.0 and float(m.critic.views['W1'][:, 10:].abs().max()) == 0.0
    obs = {'low_dim': {'flat_inputs': torch.zeros(3, case['shape']['D'])}}
    cells = (torch.zeros(1, 3, 10), torch.zeros(1, 3, 10))
    pd, new = m.forward_actor_expose_cells(obs, cells)
    assert tuple(new[0].shape) == (1, 3, 10) and tuple(pd.shape) == (3, 2 * case['shape']['A'])
