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
