"""Shared body of the rollout-worker LOOP parity tests (tests only): ``surreal_amd.agent.PPOAgent``
driven through ``main_setup`` / ``main_loop`` against tests/golden/agent_loop.json, which
oracle/gen_golden_agentloop.py recorded from the REFERENCE's own ``Agent.main_setup / main_loop``,
hooks, ``PeriodicTracker``, ``ParameterClient`` and monitors (surreal/agent/base.py:160-271,
355-363; session/tracker.py:10-45; parameter_server.py:219-303; env/monitor.py:114-218) over the
scripted environment of tests/env_fakes.py and a scripted parameter server.

Checked per case, in this order: the exact sequence of hook / fetch / request / act / env calls;
the server requests (``parameter:<last hash>``) and replies; every counter after every episode
(episodes, steps, per-parameter-update counters and their moving averages, the fetch tracker); the
scalars that reached the agent's and the environment monitor's sinks; every step's observation,
action and policy distribution (1e-5: this pins WHICH parameter version was in force at each step);
the experience windows handed to the sender.  The CPU tier runs the policy on the torch-CPU kernel
double, the GPU tier (-m gpu) on the HIP kernels through the C ABI."""
import json
import os

import numpy as np

import env_fakes as F
import helpers as H
from surreal_amd import synthetic
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config

G = json.load(open(os.path.join(H.GOLDEN_DIR, 'agent_loop.json')))
META = G['meta']
CASES = list(G['cases'])
ATOL = RTOL = 1e-5


class FakeClock(object):
    now = 1000.0

    def time(self):
        return self.now

    def sleep(self, s):
        self.now += s


class ScriptedServer(object):
    """the same script as the generator's: version v is published once the env has taken
    publish_at[v - 1] steps"""

    def __init__(self, env_steps, publish_at, blobs, log):
        self.env_steps, self.publish_at, self.blobs, self.log = env_steps, publish_at, blobs, log
        self.requests = []

    def version(self):
        return sum(1 for s in self.publish_at if s <= self.env_steps())

    def request(self, req):
        v = self.version()
        info = {'time': 990.0 + v, 'iteration': 10 * v, 'message': 'v%d' % v, 'hash': 'hash%d' % v}
        if v == 0:
            reply, what = (None, None), 'none'
        elif ':' in req and req.split(':', 1)[1] == info['hash']:
            reply, what = (None, info), 'unchanged'
        else:
            reply, what = (self.blobs[v], info), 'v%d' % v
        self.requests.append([req, what])
        self.log.append('request %s -> %s' % (req, what))
        return reply


class ScalarCapture(object):
    def __init__(self):
        self.calls = []

    def add_scalars(self, scalars, global_step=None):
        self.calls.append([sorted([k, float(v)] for k, v in scalars.items() if k != 'step_per_s'), global_step])


def _wrap_logged(obj, name, log, fmt=None):
    inner = getattr(obj, name)

    def logged(*a, **k):
        log.append(name if fmt is None else fmt(*a, **k))
        return inner(*a, **k)
    setattr(obj, name, logged)


def check_case(name, monkeypatch):
    import surreal_amd.agent.base as AB
    import surreal_amd.env.monitor as MON
    from surreal_amd.agent import PPOAgent
    from surreal_amd.distributed import ModuleDict, ParameterClient
    from surreal_amd.env import Env
    g = G['cases'][name]
    c = g['case']
    D, A = META['D'], META['A']
    clock = FakeClock()
    monkeypatch.setattr(AB, 'time', clock)
    monkeypatch.setattr(MON, 'time', clock)
    lc = ppo_learner_config()
    lc.model.actor_fc_hidden_sizes = lc.model.critic_fc_hidden_sizes = list(META['hidden'])
    lc.algo.rnn.if_rnn_policy = False
    lc.algo.n_step, lc.algo.stride = META['n_step'], META['stride']
    ec = ppo_env_config(D, A)
    ec.limit_episode_length = c['limit']
    ec.stochastic_eval = True
    sc = ppo_session_config('/tmp/surreal_amd_test_agent_loop')
    sc.agent.fetch_parameter_mode, sc.agent.fetch_parameter_interval = c['fetch_mode'], c['interval']
    sc.tensorplex.update_schedule.agent = 2
    sc.tensorplex.update_schedule.training_env = 2
    sc.tensorplex.update_schedule.eval_env = c.get('eval_env', 2)
    sc.tensorplex.update_schedule.eval_env_sleep = 0
    params = [synthetic.make_ppo_params(D, A, hidden=tuple(META['hidden']), seed=s, **META['param_kw'])
              for s in META['param_seeds']]
    zstates = [synthetic.make_zfilter_state(D, seed=5 + v) for v in range(len(params))]
    log = []
    np.random.seed(100)
    ag = PPOAgent(lc, ec, sc, agent_id=0, agent_mode=c['mode'])
    assert ag.agent_mode == g['resolved_mode'] and float(ag.noise) == g['noise']
    blobs = {}
    for v in range(1, len(params)):
        ag.model.load_params(params[v])
        ag.model.z_filter.load_state_dict(zstates[v])
        blobs[v] = ModuleDict({'ppo': ag.model}).dumps()
    ag.model.load_params(params[0])
    ag.model.z_filter.load_state_dict(zstates[0])

    env0 = F.make_scripted_loop_env(Env)(D, A, c['lens'], seed=META['env_seed'])
    ag.set_env_factory(lambda: env0)
    server = ScriptedServer(lambda: env0.total_steps, c['publish_at'], blobs, log)
    ag.attach_parameter_client(ParameterClient(server.request))
    agent_scalars = ScalarCapture()
    ag.tensorplex.sink = agent_scalars
    windows = []
    ag.set_experience_sink(lambda exp: windows.append(F.to_plain(exp)))
    for h in ('pre_episode', 'pre_action', 'post_episode', 'fetch_parameter', 'on_parameter_fetched'):
        _wrap_logged(ag, h, log)
    _wrap_logged(ag, 'post_action', log, lambda o, a, on, r, d, i: 'post_action done=%s' % bool(d))

    steps = iter(g['steps'])
    inner_act = ag.act
    seen = []

    def act(obs):
        log.append('act')
        want = next(steps)
        np.testing.assert_array_equal(np.asarray(obs['low_dim']['flat_inputs']), np.float32(want['obs']))
        if want['eps'] is not None:
            # the reference's draw for this step, injected: DiagGauss.sample reads numpy's global stream
            st = np.random.get_state()
            check = np.random.randn(1, A)
            np.random.set_state(st)
            np.testing.assert_array_equal(check[0], np.float64(want['eps']))
        got = inner_act(obs)
        a, info = got if ag.agent_mode == 'training' else (got, None)
        step = len(seen)
        np.testing.assert_allclose(np.asarray(a), want['action'], atol=ATOL, rtol=RTOL,
                                   err_msg='%s: action of step %d' % (name, step))
        if info is not None:
            np.testing.assert_allclose(np.asarray(info[1][0]), want['pd'], atol=ATOL, rtol=RTOL,
                                       err_msg='%s: policy distribution of step %d' % (name, step))
            assert info[0] == []
        seen.append(a)
        return got
    ag.act = act

    np.random.seed(300)
    ag.main_setup()
    env_scalars = ScalarCapture()
    probe = ag.env
    while probe is not None:
        if type(probe).__name__.endswith('TensorplexMonitor'):
            probe.tensorplex = env_scalars
        probe = getattr(probe, 'env', None)
    _wrap_logged(env0, '_reset', log, lambda: 'env.reset')
    _wrap_logged(env0, '_step', log, lambda a: 'env.step')
    log.append('setup done')
    for ep in range(c['episodes']):
        ag.main_loop()
        log.append('episode done')
        want = g['episodes'][ep]
        tr = ag._fetch_parameter_tracker
        got = dict(current_episode=ag.current_episode, cumulative_steps=ag.cumulative_steps,
                   current_step=ag.current_step, actions_since_param_update=ag.actions_since_param_update,
                   episodes_since_param_update=ag.episodes_since_param_update, env_total_steps=env0.total_steps,
                   windows=len(windows), tracker_value=tr.value, tracker_endpoint=tr._endpoint)
        for k, v in got.items():
            assert v == want[k], '%s episode %d: %s = %r, the reference has %r' % (name, ep, k, v, want[k])
        if c['mode'] == 'training':
            for k in ('actions_per_param_update', 'episodes_per_param_update'):
                np.testing.assert_allclose(float(getattr(ag, k).cur_value()), want[k], rtol=1e-12, err_msg=k)
    # ---- the order of everything that happened ---------------------------------------------------
    assert log == g['log'], '%s: hook / fetch order differs from the reference at entry %d' % (
        name, next(i for i, (x, y) in enumerate(zip(log + [None], g['log'] + [None])) if x != y))
    assert server.requests == g['requests']
    assert len(seen) == len(g['steps'])

    def scalars_close(got_calls, want_calls, what):
        assert len(got_calls) == len(want_calls), what
        for (gs, gstep), (ws, wstep) in zip(got_calls, want_calls):
            assert gstep == wstep and [k for k, _ in gs] == [k for k, _ in ws], what
            np.testing.assert_allclose([v for _, v in gs], [v for _, v in ws], rtol=1e-9, atol=1e-12, err_msg=what)
    scalars_close(agent_scalars.calls, g['agent_scalars'], name + ': agent scalars')
    scalars_close(env_scalars.calls, g['env_scalars'], name + ': env monitor scalars')
    # ---- the experience windows (exp_sender_wrapper.py:153-264 fed by this loop) -----------------
    assert len(windows) == len(g['windows'])
    for i, (got_w, want_w) in enumerate(zip(windows, g['windows'])):
        H.assert_plain_close(got_w, want_w, ATOL, RTOL, '%s: window %d' % (name, i))
