"""CPU tier: DDPGLearner host logic (update order, target-update schedule, clipping switches,
Adam step counters) through the CPU kernel double against goldens recorded from the reference's
own DDPGLearner (oracle/gen_golden_ddpg.py), and the oracle restatement against the same."""
import json

import numpy as np
import pytest

import ddpg_helpers as DH
import ddpg_oracle
from surreal_amd import synthetic


@pytest.mark.parametrize('name', DH.DDPG_CASES)
def test_ddpg_oracle_matches_reference_golden(name):
    g, c = DH.load(name)
    h = c['hyper']
    pixel = tuple(c['pixel']) if c.get('pixel') else None

    def mkp(seed):
        if pixel is not None:
            return ddpg_oracle.make_ddpg_pixel_params(c['D'], c['A'], pixel, c['conv_hidden'], tuple(c['ah']),
                                                      tuple(c['ch']), seed=seed, layernorm=bool(h.get('layernorm', False)))
        return ddpg_oracle.make_ddpg_params(c['D'], c['A'], tuple(c['ah']), tuple(c['ch']), seed=seed,
                                            layernorm=bool(h.get('layernorm', False)))
    params, params2 = mkp(3), mkp(4)
    O = ddpg_oracle.OracleDDPGLearner(
        params, gamma=h['gamma'], n_step=h['n_step'], lr_actor=h['lr_actor'], lr_critic=h['lr_critic'],
        clip_critic_gradient=h.get('clip_critic', False), target_update_type=h['target_update_type'],
        target_update_interval=h['target_update_interval'], tau=h.get('tau', 1e-3),
        use_double_critic=h.get('double_critic', False), use_action_regularization=h.get('action_reg', False),
        params2=params2, batch_size=c['B'])
    ref = json.loads(str(g['trace_json']))
    for it in range(c['iters']):
        np.random.seed(1000 + it)
        st = O.learn(synthetic.make_ddpg_batch(c['B'], c['D'], c['A'], seed=10 + it, pixel=pixel))
        assert set(st) == set(ref[it])
        for k, v in ref[it].items():
            np.testing.assert_allclose(st[k], v, atol=2e-6, rtol=2e-6)


@pytest.mark.parametrize('name', DH.DDPG_CASES)
def test_ddpg_learner_host_logic(name, cpu_double):
    DH.run_and_check(name)


@pytest.mark.parametrize('name', ['tiny_hard', 'tiny_soft_clipcritic', 'cfg3_cheetah512'])
def test_ddpg_learner_host_logic_both_schedules(name, cpu_double):
    """the row-block schedule's host side (the default up to 1024 rows): which buffers the two chain launches and the
    weight-gradient launches share, and WHEN the packed weight copy is refreshed -- the double works from a snapshot of
    the parameters taken by ddpg_rows_pack, so a missing refresh (the critic's, between its Adam step and the actor
    phase) fails the goldens here as it would on the device"""
    L = DH.run_and_check(name)
    assert getattr(L._ws, 'rows_args', None) is not None
    L = DH.run_and_check(name, opts={'ddpg_row_schedule': False})          # the level schedule
    assert getattr(L._ws, 'rows_args', None) is None


def test_replay_samples_straight_into_the_learners_staging_buffers(cpu_double):
    DH.check_sampling_into_staging('cpu')


def test_ddpg_every_switch_combination_constructs_and_learns(cpu_double):
    """use_layernorm x double critic x camera observations (round 6: none refuses any more; the goldens tiny_ln_td3_soft,
    tiny_ln_pixel_hard, tiny_ln_pixel_td3_soft pin the arithmetic); an unknown target-update type is a ConfigError"""
    from surreal_amd.learner.ddpg import DDPGLearner
    from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config
    for double, pixel in ((True, None), (False, (2, 20, 24)), (True, (2, 20, 24))):
        lc = ddpg_learner_config()
        lc.model.use_layernorm = True
        lc.model.actor_fc_hidden_sizes, lc.model.critic_fc_hidden_sizes = [24, 16], [32, 24]
        lc.algo.network.use_double_critic = double
        lc.replay.batch_size = 8
        if pixel:
            lc.model.conv_spec.hidden_output_dim = 8
        L = DDPGLearner(lc, ddpg_env_config(5, 2, pixel=pixel), ddpg_session_config())
        st = dict(L.learn(synthetic.make_ddpg_batch(8, 5, 2, seed=1, pixel=pixel)))
        assert np.isfinite(st['critic_loss']) and ('Q_policy2' in st) == double
    lc = ddpg_learner_config()
    lc.algo.network.target_update = {'type': 'weird'}
    from surreal_amd.session import ConfigError
    with pytest.raises(ConfigError):
        DDPGLearner(lc, ddpg_env_config(5, 2), ddpg_session_config())


def test_ddpg_agent_replay_learner_loop(cpu_double):
    """DDPG in-process loop: agent -> n-step SSAR wrapper -> uniform replay -> learner"""
    import torch
    from surreal_amd.agent import DDPGAgent
    from surreal_amd.env import SyntheticEnv, MaxStepWrapper
    from surreal_amd.learner import DDPGLearner
    from surreal_amd.replay import UniformReplay
    from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config
    D, A = 7, 2
    lc = ddpg_learner_config()
    lc.model.actor_fc_hidden_sizes, lc.model.critic_fc_hidden_sizes = [24, 16], [32, 24]
    lc.replay.batch_size, lc.replay.memory_size, lc.replay.sampling_start_size = 8, 64, 10
    ec, sc = ddpg_env_config(D, A, num_agents=2), ddpg_session_config()
    ec.limit_episode_length = 15
    replay = UniformReplay(lc, ec, sc)
    learner = DDPGLearner(lc, ec, sc)
    learner.attach_replay(replay)
    ag = DDPGAgent(lc, ec, sc, agent_id=1, agent_mode='training')
    assert ag.sigma == 0.5
    ag.set_experience_sink(replay._insert_wrapper)
    ag.set_env_factory(lambda: SyntheticEnv(D, A, episode_len=100, seed=4))
    ag.attach_learner(learner)
    ag.main_setup()
    ag.main_loop()
    assert len(replay) == 15 - (lc.algo.n_step - 1)      # the last n_step-1 transitions are never sent
    assert replay.start_sample_condition()
    learner.main_setup()
    learner.main_loop()
    st = learner.tensorplex.latest
    assert all(np.isfinite(st[k]) for k in ('actor_loss', 'critic_loss', 'Q_target', 'Q_policy'))
    acts = ag.act_batch(torch.randn(5, D), eps=torch.zeros(5, A))
    assert float(acts.abs().max()) <= 1.0


def test_ddpg_pixel_agent_acts_on_camera_frames(cpu_double):
    """DDPGAgent.act with pixel observations: perception CNN on camera0 / 255 in front of the actor
    (ddpg_agent.py:155-184, ddpg_net.py:67-88), parameters fetched from a pixel learner"""
    import collections
    import torch
    from surreal_amd.agent import DDPGAgent
    from surreal_amd.learner import DDPGLearner
    from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config
    g, c = DH.load('tiny_pixel_hard')
    L = DH.make_learner(c)
    lc = L.learner_config
    ec, sc = ddpg_env_config(c['D'], c['A'], pixel=tuple(c['pixel'])), ddpg_session_config()
    ag = DDPGAgent(lc, ec, sc, agent_id=0, agent_mode='eval_deterministic_local')
    ag.attach_learner(L)
    assert ag.fetch_parameter() and torch.equal(ag.model.perception_flat, L.model.perception_flat)
    b = synthetic.make_ddpg_batch(3, c['D'], c['A'], seed=5, pixel=tuple(c['pixel']))
    obs = collections.OrderedDict(pixel={'camera0': b['obs']['pixel']['camera0'][1]},
                                  low_dim={'flat_inputs': b['obs']['low_dim']['flat_inputs'][1]})
    a = ag.act(obs)
    params = ddpg_oracle.make_ddpg_pixel_params(c['D'], c['A'], tuple(c['pixel']), c['conv_hidden'],
                                                tuple(c['ah']), tuple(c['ch']), seed=3)
    O = ddpg_oracle.OracleDDPGModel(params)
    t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32)[None]  # noqa: E731
    want = O.forward_actor(O.forward_perception({'pixel': {'camera0': t(obs['pixel']['camera0'])},
                                                 'low_dim': {'flat_inputs': t(obs['low_dim']['flat_inputs'])}}))
    np.testing.assert_allclose(a, want.detach().numpy()[0].clip(-1, 1), atol=1e-6)


def test_frame_stack_preprocessor_joins_frame_lists(cpu_double):
    """frame stacks shipped as lists of frames (frame_stack_concatenate_on_env off) are joined on the
    channel axis before SSAR aggregation (aggregator.py:11-30, ddpg.py:430-440)"""
    import collections
    from surreal_amd.learner import DDPGLearner
    from surreal_amd.learner.aggregator import FrameStackPreprocessor
    from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config

    def obs(t):
        frames = [np.full((1, 20, 24), t + k, dtype=np.uint8) for k in range(2)]
        return collections.OrderedDict(pixel={'camera0': frames}, low_dim={'flat_inputs': np.full(3, t, np.float32)})
    exps = [{'obs': [obs(i), obs(i + 1)], 'action': np.zeros(2, np.float32), 'reward': 1.0, 'done': False}
            for i in range(4)]
    lc = ddpg_learner_config()
    lc.model.actor_fc_hidden_sizes, lc.model.critic_fc_hidden_sizes = [12, 8], [16, 12]
    lc.model.conv_spec.hidden_output_dim = 8
    lc.replay.batch_size = 4
    ec = ddpg_env_config(3, 2, pixel=(2, 20, 24))
    ec.frame_stack_concatenate_on_env = False
    L = DDPGLearner(lc, ec, ddpg_session_config())
    b = L._prefetcher_preprocess(exps)
    assert b['obs']['pixel']['camera0'].shape == (4, 2, 20, 24) and b['obs']['pixel']['camera0'].dtype == np.uint8
    assert b['obs_next']['pixel']['camera0'][2, 1, 0, 0] == 4 and b['obs']['low_dim']['flat_inputs'].shape == (4, 3)
    st = L.learn(b)                                            # and the pixel learner takes it
    assert np.isfinite(st['critic_loss'])
    # the replay's own objects are left as the agent sent them: neighbouring SSAR experiences SHARE
    # an observation dict (exp_t's obs[1] is exp_{t+1}'s obs[0]) and uniform sampling repeats
    # experiences, so a second pass over the same objects must see frame lists again
    shared = [obs(i) for i in range(5)]
    exps2 = [{'obs': [shared[i], shared[i + 1]], 'action': np.zeros(2, np.float32), 'reward': 1.0, 'done': False}
             for i in range(4)]
    b1 = L._prefetcher_preprocess([exps2[0], exps2[1], exps2[1], exps2[3]])
    b2 = L._prefetcher_preprocess([exps2[0], exps2[1], exps2[1], exps2[3]])
    np.testing.assert_array_equal(b1['obs']['pixel']['camera0'], b2['obs']['pixel']['camera0'])
    assert all(isinstance(o['pixel']['camera0'], list) for o in shared)
    with pytest.raises(AssertionError):
        FrameStackPreprocessor.preprocess_obs({'pixel': {'camera0': [np.zeros((2, 3))]}})


def test_resume_across_the_hard_update_at_configs2_size(cpu_double):
    DH.check_resume_across_hard_update()
