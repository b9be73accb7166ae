"""GPU tier: DDPGLearner on the HIP path against the reference goldens (1e-5)."""
import pytest

import ddpg_helpers as DH

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', DH.DDPG_CASES)
def test_ddpg_learner_matches_reference_golden(name):
    DH.run_and_check(name)


def test_ddpg_resume_across_the_hard_update_at_configs2_size():
    """ddpg.py:403-428 at interval 500 and batch 512, through the captured graph, at the 1e-5 bar (see the helper)"""
    DH.check_resume_across_hard_update()


ROWS_CASES = ['tiny_hard', 'tiny_soft_clipcritic', 'cfg3_cheetah512']     # low-dimensional, one critic, no LayerNorm


@pytest.mark.parametrize('name', ROWS_CASES)
def test_ddpg_learner_matches_reference_golden_both_schedules(name):
    """the reference goldens through the row-block launches (the default up to 1024 rows) and through the level schedule
    (session_config.learner.ddpg_row_schedule = False)"""
    L = DH.run_and_check(name)
    assert getattr(L._ws, 'rows_args', None) is not None
    L = DH.run_and_check(name, opts={'ddpg_row_schedule': False})
    assert getattr(L._ws, 'rows_args', None) is None
    # ... and with the weight gradients as their own launch in front of the update launch (what several ranks run)
    L = DH.run_and_check(name, opts={'ddpg_rows_fused_update': False})
    assert getattr(L._ws, 'rows_args', None) is not None


@pytest.mark.parametrize('name', ROWS_CASES)
def test_row_schedule_agrees_with_the_level_schedule(name):
    """smx_ddpg_rows.hip sums a layer's products in the MFMA loop's order, smx_linear_f32 in its own: after several
    iterations parameters, targets and statistics agree within fp32 rounding of the level schedule's (measured: see the
    assertion messages' bounds), not bit for bit"""
    import copy
    import numpy as np
    import torch
    from surreal_amd import synthetic
    g, case = DH.load(name)
    rows, levels = DH.make_learner(case, {'ddpg_row_schedule': True}), DH.make_learner(case, {'ddpg_row_schedule': False})
    for it in range(4):
        b = synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it)
        sa, sb = dict(rows.learn(copy.deepcopy(b))), dict(levels.learn(copy.deepcopy(b)))
        for k in sb:
            np.testing.assert_allclose(sa[k], sb[k], rtol=2e-5, atol=2e-6, err_msg='%s iteration %d' % (k, it))
    assert getattr(rows._ws, 'rows_args', None) is not None and getattr(levels._ws, 'rows_args', None) is None
    lr = max(case['hyper']['lr_actor'], case['hyper']['lr_critic'])
    for a, b in ((rows.model, levels.model), (rows.model_target, levels.model_target)):
        for x, y in ((a.actor_flat, b.actor_flat), (a.critic_flat, b.critic_flat)):
            d = (x - y).abs()
            # (Adam's first steps move a weight by ~lr sign(g): an element whose gradient sits at the rounding noise
            # floor may differ by that much, the rest agree to ~1e-7)
            assert float(d.max()) <= 2 * lr * 4 + 1e-6, float(d.max())
            assert float((d > 2e-6).float().mean()) < 0.02, float((d > 2e-6).float().mean())


@pytest.mark.parametrize('name,B', [('tiny_hard', None), ('tiny_hard', 37), ('cfg3_cheetah512', None), ('cfg3_cheetah512', 1000), ('tiny_hard', 1030),
                                    ('cfg3_cheetah512', 1100)])
def test_row_block_launches_fill_the_level_schedules_buffers(name, B):
    """one iteration, buffer by buffer: what the two row-block launches leave in the workspace (activations, Bellman
    targets, data gradients of both networks) and the weight gradients formed from it, against the level schedule's --
    also with batches that are not multiples of the row block (4 rows; past 1024 rows -- where the learner would pick the level schedule by itself -- several rounds of workgroups)"""
    import copy
    import torch
    from surreal_amd import synthetic
    g, case = DH.load(name)
    B = B or case['B']
    rows, levels = DH.make_learner(case, {'ddpg_row_schedule': True}), DH.make_learner(case, {'ddpg_row_schedule': False})
    b = synthetic.make_ddpg_batch(B, case['D'], case['A'], seed=77)
    rows.learn(copy.deepcopy(b))
    levels.learn(copy.deepcopy(b))
    torch.cuda.synchronize()
    wr, wl = rows._ws, levels._ws
    c1 = rows.model.c1
    pairs = [(k, getattr(wr, k), getattr(wl, k)) for k in ('xcat', 'h2c', 'q', 'q_next', 'y', 'dz3', 'h1a', 'h2a', 'act',
                                                           'q_actor', 'dz3a', 'dz2a', 'dz1a', 'grads_c', 'grads_a')]
    pairs.append(('dz1 (critic)', wr.dxcat[:, :c1], wl.dxcat[:, :c1]))
    for k, x, y in pairs:
        scale = float(y.abs().max()) + 1e-30
        d = float((x - y).abs().max())
        assert d <= 2e-6 * max(scale, 1.0) + 1e-5 * scale, '%s: max |diff| %g at scale %g' % (k, d, scale)
    assert int(wr.step[0]) == int(wl.step[0])


@pytest.mark.parametrize('name', ['tiny_hard', 'tiny_soft_clipcritic', 'cfg3_cheetah512'])
def test_level_schedule_equals_layer_schedule_bit_for_bit(name):
    """the dependency-level schedule (independent layers of the four forward chains and a level's weight gradients
    share launches: smx_linear_multi_f32) runs the same kernels on the same operands as one launch per layer: after
    several iterations the parameters, target parameters and reported statistics are identical to the last bit"""
    import copy
    import torch
    from surreal_amd import synthetic
    g, case = DH.load(name)
    learners = []
    for levels in (True, False):
        L = DH.make_learner(case, {'ddpg_row_schedule': False})
        L.level_schedule = levels
        learners.append(L)
    assert not (learners[0].is_pixel_input or learners[0].use_double_critic)
    for it in range(4):
        b = synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it)
        sa = dict(learners[0].learn(copy.deepcopy(b)))
        sb = dict(learners[1].learn(copy.deepcopy(b)))
        assert sa == sb, (it, sa, sb)
    for a, b in ((learners[0].model, learners[1].model), (learners[0].model_target, learners[1].model_target)):
        assert torch.equal(a.actor_flat, b.actor_flat) and torch.equal(a.critic_flat, b.critic_flat)


def test_ddpg_stats_against_float64_and_its_nan_check():
    """the statistics launch (ddpg.py:335-342): means in float64 order-independent to 1e-6, max |a| exact -- and NaN
    whenever ANY action is NaN, whatever comes behind it in the row (the |a| <= 1 assertion of ddpg.py:262-263 must not
    pass on a NaN action)"""
    import numpy as np
    import torch
    from surreal_amd.kernels import HipKernels
    K = HipKernels()
    g = torch.Generator(device='cuda').manual_seed(5)
    for rows, A in ((512, 6), (1000, 3), (37, 17)):
        q, y, r, qa = (torch.randn(rows, generator=g, device='cuda') for _ in range(4))
        act = torch.rand(rows, A, generator=g, device='cuda') * 2 - 1
        st = torch.zeros(8, device='cuda')
        K.ddpg_stats(q, y, r, act, qa, st)
        want = [-qa.double().mean(), ((q - y).double() ** 2).mean(), act.double().norm(2, 1).mean(), r.double().mean(),
                y.double().mean(), q.double().mean()]
        np.testing.assert_allclose(st[:6].cpu().numpy(), [float(v) for v in want], rtol=2e-6, atol=1e-7)
        assert float(st[6]) == float(act.abs().max())
        for (i, j) in ((0, 0), (rows - 1, A - 1), (rows // 2, 0)):
            bad = act.clone()
            bad[i, j] = float('nan')
            K.ddpg_stats(q, y, r, bad, qa, st)
            assert np.isnan(float(st[6])), (rows, A, i, j)


def test_replay_samples_straight_into_the_learners_staging_buffers():
    DH.check_sampling_into_staging('cuda')


@pytest.mark.parametrize('name', ['tiny_hard', 'tiny_soft_clipcritic', 'cfg3_cheetah512'])
def test_update_launch_keeps_the_packed_copies_current(name):
    """smx_ddpg_rows_update_f32 writes every updated weight (model and target) into the fragment-order copies: after
    several iterations -- across a hard update where the case has one -- the packed buffer is, bit for bit, what a full
    smx_ddpg_rows_pack_f32 of the parameters gives"""
    import copy
    import torch
    from surreal_amd import synthetic
    g, case = DH.load(name)
    L = DH.make_learner(case, {'ddpg_row_schedule': True})
    for it in range(max(5, case['hyper'].get('target_update_interval', 1) + 2) if case['hyper']['target_update_type'] == 'hard' and
                    case['hyper']['target_update_interval'] <= 10 else 5):
        L.learn(copy.deepcopy(synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it)))
    torch.cuda.synchronize()
    ws = L._ws
    kept = ws.rows_packed.clone()
    L.K.ddpg_rows_pack(ws.rows_args)
    torch.cuda.synchronize()
    assert torch.equal(kept, ws.rows_packed)
    assert float(kept.abs().sum()) > 0


def test_parameters_written_from_outside_are_repacked():
    """a state dict loaded between iterations (torch writes: the buffers' version counters move) reaches the row
    kernels: the next iteration equals the level schedule's from the same parameters"""
    import copy
    import numpy as np
    import torch
    from surreal_amd import synthetic
    g, case = DH.load('cfg3_cheetah512')
    rows, levels = DH.make_learner(case, {'ddpg_row_schedule': True}), DH.make_learner(case, {'ddpg_row_schedule': False})
    mk = lambda it: copy.deepcopy(synthetic.make_ddpg_batch(case['B'], case['D'], case['A'], seed=10 + it))  # noqa: E731
    for it in range(3):
        rows.learn(mk(it)); levels.learn(mk(it))
    gen = torch.Generator(device='cuda').manual_seed(5)
    for L in (rows, levels):
        pass
    newc = torch.randn(rows.model.critic_flat.shape, generator=gen, device='cuda') * 0.05
    newa = torch.randn(rows.model.actor_flat.shape, generator=gen, device='cuda') * 0.05
    for L in (rows, levels):
        L.model.critic_flat.copy_(newc)
        L.model_target.actor_flat.copy_(newa)
    sa, sb = dict(rows.learn(mk(3))), dict(levels.learn(mk(3)))
    for k in sb:
        np.testing.assert_allclose(sa[k], sb[k], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('D,A,ah,ch,B', [
    (1, 1, (4, 4), (4, 4), 5),                    # the smallest shapes the row kernels take
    (50, 32, (1024, 64), (64, 1024), 130),        # two head tiles; K = 1024 split over the waves in two trips; 33 blocks
    (300, 17, (300, 200), (400, 300), 1024),      # the largest batch on 4-row blocks
    (2048, 3, (128, 36), (36, 128), 64),          # the widest observation
    (17, 6, (304, 204), (404, 300), 515),         # tile counts off the multiples of eight, a ragged last block
])
def test_row_blocks_shape_sweep_against_the_level_schedule(D, A, ah, ch, B):
    """shapes around the row kernels' branches (split-K heads, passes of 32 feature tiles, chunk counts 4 j and 4 j + 2,
    ragged last blocks): one iteration's buffers against the level schedule's, as in
    test_row_block_launches_fill_the_level_schedules_buffers"""
    import copy
    import torch
    from surreal_amd import synthetic
    g, case = DH.load('tiny_soft_clipcritic')
    # (soft target update; the learning rates of configs[2] -- at the tiny cases' 1e-2 an Adam step of the critic on a
    # noise-floor gradient moves the ACTOR phase of the same iteration by per cents: ReLU masks of Q(s, mu(s)) flip)
    case = dict(case, D=D, A=A, ah=list(ah), ch=list(ch), B=B, hyper=dict(case['hyper'], lr_actor=1e-4, lr_critic=1e-3))
    rows, levels = DH.make_learner(case, {'ddpg_row_schedule': True}), DH.make_learner(case, {'ddpg_row_schedule': False})
    b = synthetic.make_ddpg_batch(B, D, A, seed=91)
    rows.learn(copy.deepcopy(b))
    levels.learn(copy.deepcopy(b))
    torch.cuda.synchronize()
    wr, wl = rows._ws, levels._ws
    assert getattr(wr, 'rows_args', None) is not None and getattr(wl, 'rows_args', None) is None
    c1 = rows.model.c1
    pairs = [(k, getattr(wr, k), getattr(wl, k)) for k in ('xcat', 'h2c', 'q', 'q_next', 'y', 'dz3', 'h1a', 'h2a', 'act',
                                                           'q_actor', 'dz3a', 'dz2a', 'dz1a', 'grads_c', 'grads_a')]
    pairs.append(('dz1 (critic)', wr.dxcat[:, :c1], wl.dxcat[:, :c1]))
    for k, x, y in pairs:
        scale = float(y.abs().max()) + 1e-30
        d = float((x - y).abs().max())
        assert d <= 2e-6 * max(scale, 1.0) + 2e-5 * scale, '%s: max |diff| %g at scale %g' % (k, d, scale)
    rows.learn(copy.deepcopy(synthetic.make_ddpg_batch(B, D, A, seed=92)))     # (the update launches' copies, twice)
    torch.cuda.synchronize()
    kept = wr.rows_packed.clone()
    rows.K.ddpg_rows_pack(wr.rows_args)
    torch.cuda.synchronize()
    assert torch.equal(kept, wr.rows_packed)
