"""GPU tier: DDPGLearner on the HIP path against the reference goldens (1e-5)."""
import pytest

import ddpg_helpers as DH

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', DH.DDPG_CASES)
def test_ddpg_learner_matches_reference_golden(name):
    DH.run_and_check(name)


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
        L = DH.make_learner(case)
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
