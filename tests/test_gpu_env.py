"""GPU tier (-m gpu), SURVEY.md 8(f) rank 4: a rollout worker whose policy runs on the HIP path acting THROUGH
the env adapter / observation-transform / monitor stack of ``make_env`` + ``Agent.prepare_env``
(surreal/env/wrapper.py:165-513, make_env.py:93-104, monitor.py:114-218, agent/base.py:283-336).

  * the observations the wrapper stack hands the agent are the ones tests/golden/envwrap.json recorded from the
    REFERENCE's own wrappers on the same scripted simulator (tests/env_fakes.py), step for step;
  * what the agent answers is the ORACLE model's policy on those observations (oracle/ppo_oracle.py, fp32 ATen:
    [mean | std * exp(noise)], sample with the same normal draws, clip) at 1e-5 -- low-dim MLP policy, and camera
    frames (uint8, channel-first after TransposeWrapper, grayscale + frame stack) through the CNN stem on the GPU;
  * the environment monitor's scalars and the emitted experience windows follow from those actions.
"""
import collections
import json
import os

import numpy as np
import pytest
import torch

import env_fakes as F
import helpers as H
import ppo_oracle
from surreal_amd import env as E
from surreal_amd import synthetic
from surreal_amd.session import Config

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(H.GOLDEN_DIR, 'envwrap.json')))


def plain(x):
    return json.loads(json.dumps(F.to_plain(x)))


def robosuite_cfg(**kw):
    base = dict(pixel_input=True, use_depth=False, use_grayscale=False, frame_stacks=0,
                frame_stack_concatenate_on_env=True, action_repeat=1,
                observation={'pixel': ['camera0'], 'low_dim': ['robot-state', 'object-state']})
    base.update(kw)
    return Config(base)


def _agent(env, cfg, hidden, pixel, seed):
    from surreal_amd.agent import PPOAgent
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    D = int(sum(v[0] for v in cfg.obs_spec['low_dim'].values()))
    A = int(cfg.action_spec['dim'][0])
    cam = tuple(cfg.obs_spec['pixel']['camera0']) if pixel else None
    lc = ppo_learner_config()
    lc.algo.rnn.if_rnn_policy = False
    lc.algo.n_step, lc.algo.stride = 3, 2
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = list(hidden)
    lc.model.cnn_feature_dim = 8
    ec = ppo_env_config(D, A, pixel=cam)
    ec.obs_spec, ec.action_spec = cfg.obs_spec, cfg.action_spec       # what make_env resolved
    ec.limit_episode_length = 4
    sc = ppo_session_config('/tmp/surreal_amd_test_gpu_env')
    sc.tensorplex.update_schedule.training_env = 2
    np.random.seed(seed)
    ag = PPOAgent(lc, ec, sc, agent_id=3, agent_mode='training')
    assert ag.model.flat.is_cuda
    pix_kw = dict(pixel=cam, cnn_feature_dim=8) if pixel else {}
    params = synthetic.make_ppo_params(D, A, hidden=tuple(hidden), seed=41, final_scale=2.0, log_sig_spread=0.4, **pix_kw)
    zstate = synthetic.make_zfilter_state(D, seed=6)
    ag.model.load_params(params)
    ag.model.z_filter.load_state_dict(zstate)
    oracle = ppo_oracle.OraclePPOModel(params, A, True, zstate, in_size=D)
    ag.set_env_factory(lambda: env)
    return ag, oracle, A


def _run(ag, oracle, A, episodes):
    """main_setup + `episodes` main_loops; every act() is checked against the oracle policy on the same observation"""
    windows, seen = [], []
    ag.set_experience_sink(windows.append)
    inner = ag.act

    def act(obs):
        st = np.random.get_state()
        eps = np.random.randn(1, A)
        np.random.set_state(st)
        a, info = inner(obs)
        t = {m: {k: torch.tensor(np.asarray(v), dtype=torch.float32).unsqueeze(0) for k, v in d.items()}
             for m, d in obs.items()}
        with torch.no_grad():
            pd = oracle.forward_actor(t).numpy().copy()                  # ppo_agent.py:137-139
        pd[:, A:] *= np.exp(ag.noise)
        want = np.clip(eps * pd[:, A:] + pd[:, :A], -1, 1).reshape(-1)  # DiagGauss.sample + clip (:141-145)
        np.testing.assert_allclose(info[1][0], pd.reshape(-1), atol=1e-5, rtol=1e-5, err_msg='step %d pd' % len(seen))
        np.testing.assert_allclose(a, want, atol=1e-5, rtol=1e-5, err_msg='step %d action' % len(seen))
        seen.append((obs, a))
        return a, info
    ag.act = act
    np.random.seed(77)
    ag.main_setup()
    assert isinstance(ag.env, E.Wrapper)
    rewards = [ag.main_loop() for _ in range(episodes)]
    return seen, windows, rewards


def test_gpu_agent_through_the_lowdim_robosuite_stack_sees_the_reference_observations():
    cfg = robosuite_cfg(pixel_input=False, observation={'pixel': [], 'low_dim': ['object-state']})
    sim = F.FakeRobosuite(T=5)
    env, cfg = E.wrap_robosuite(sim, cfg)
    g = GOLD['robosuite']['lowdim_only']
    assert plain(env.observation_spec()) == g['obs_spec'] and plain(env.action_spec()) == g['action_spec']
    ag, oracle, A = _agent(env, cfg, (16, 12), False, seed=5)
    seen, windows, rewards = _run(ag, oracle, A, episodes=4)
    # the reference's recording of the same simulator script: reset, steps 1..5 (done), reset, ...  The step limit
    # (4) ends our episodes one step earlier; within an episode the observation of step t is the recorded one
    ref_reset = next(r['reset'] for r in g['trace'] if 'reset' in r)
    ref_steps = [r['obs'] for r in g['trace'] if 'obs' in r][:5]
    assert len(seen) == 16
    for i, (obs, a) in enumerate(seen):
        t = i % 4
        want = ref_reset if t == 0 else ref_steps[t - 1]
        assert plain(obs) == want, 'episode %d step %d observation differs from the reference wrapper stack' % (i // 4, t)
    # rewards follow the simulator's script with OUR actions (FakeRobosuite: t + sum(action)); the training monitor
    # reports the mean episode reward every 2 episodes (monitor.py:114-160)
    acts = np.array([a for _, a in seen]).reshape(4, 4, A)
    ep_rewards = [(np.arange(1, 5) + acts[e].sum(1)).sum() for e in range(4)]
    np.testing.assert_allclose(rewards, ep_rewards, rtol=1e-6)
    hist = ag.env_tensorplex.history
    assert [step for step, _ in hist] == [2, 4]
    np.testing.assert_allclose([sc[':reward'] for _, sc in hist], [np.mean(ep_rewards[:2]), np.mean(ep_rewards[2:])],
                               rtol=1e-6)
    # moving windows of n_step 3, stride 2 over 4-step episodes: one window per episode (+ the tail window rule)
    assert windows and all(len(w['obs']) == 3 and w['n_step'] == 3 for w in windows)
    assert sim.steps == 16


def test_gpu_agent_through_the_pixel_robosuite_stack_cnn_policy_equals_oracle():
    """camera frames: (H, W, 3) uint8 from the simulator -> flipped / transposed channel-first -> grayscale (uint8
    wrap-around mean) -> 3-frame stack on the channel axis -> the CNN stem on the GPU (uint8 all the way, x / 255 in
    the first convolution's gather).  The transforms are pinned to the reference on the CPU tier
    (tests/test_env_adapters.py, envwrap.json: gray_stack3 on the 6x4 simulator); here the same stack runs on a
    simulator large enough for the stem's 8x8 stride-4 convolution and the policy is held to the oracle model."""
    cfg = robosuite_cfg(use_grayscale=True, frame_stacks=3)
    small, _ = E.wrap_robosuite(F.FakeRobosuite(T=5), robosuite_cfg(use_grayscale=True, frame_stacks=3))
    assert plain(small.observation_spec()) == GOLD['robosuite']['gray_stack3']['obs_spec']
    sim = F.FakeRobosuite(T=6, H=44, W=36)
    env, cfg = E.wrap_robosuite(sim, cfg)
    assert tuple(cfg.obs_spec['pixel']['camera0']) == (3, 44, 36)
    ag, oracle, A = _agent(env, cfg, (16, 12), True, seed=6)
    assert ag.model.if_pixel
    seen, windows, rewards = _run(ag, oracle, A, episodes=2)
    assert len(seen) == 8
    for obs, _ in seen:
        fr = obs['pixel']['camera0']
        assert fr.dtype == np.uint8 and fr.shape == (3, 44, 36)
    # the stack really moves: the newest frame of step t is the oldest of step t + 2
    np.testing.assert_array_equal(seen[1][0]['pixel']['camera0'][2], seen[3][0]['pixel']['camera0'][0])
    assert windows and windows[0]['obs'][0]['pixel']['camera0'].dtype == np.uint8


def test_device_camera_and_frame_stacking_match_host_wrapper_hip():
    """"obs stacking" on the device (smx_synth_frame_u8 + smx_frame_stack_u8): raw uint8 frames rendered and stored once
    per step, the policy's stacked observation and the stacked windows by one gather -- bit-exact against SyntheticEnv
    under the host FrameStackWrapper (which tests/test_env_adapters.py pins to the reference's), incl. frames as wide as
    configs[3]'s 3 x 84 x 84 and an odd row pitch (dword / byte lanes)"""
    import pixel_env_cases as PC
    from surreal_amd import kernels as KN
    PC.check_device_camera_matches_host_framestack()
    PC.check_device_camera_matches_host_framestack(n=3, D=5, A=2, pixel=(3, 84, 84), stacks=2, T=4, n_step=2, stride=1)
    PC.check_device_camera_matches_host_framestack(n=2, D=5, A=2, pixel=(1, 9, 7), stacks=4, T=5, n_step=2, stride=2)
    PC.check_frame_stack_with_resets_inside(KN.default_kernels(), 'cuda')
