"""CPU tier: the rollout workers' host logic (mode remap, exploration-noise draws, action_info
layout, LSTM cell hand-over, episode hooks) against the fixtures recorded from the REFERENCE's own
agents (tests/golden/agents.npz), with the torch-CPU kernel double standing in for the HIP
kernels.  tests/test_gpu_agents.py runs the same checks on the HIP path."""
import pytest

import agent_cases as AC
import agent_loop_cases as AL


@pytest.mark.parametrize('name', AC.PPO_CASES)
def test_ppo_agent_matches_reference_agent(name, cpu_double):
    AC.check_ppo_case(name)


@pytest.mark.parametrize('name', AC.DDPG_CASES)
def test_ddpg_agent_matches_reference_agent(name, cpu_double):
    AC.check_ddpg_case(name)


def test_action_noise_streams_match_reference():
    AC.check_noise_streams()


@pytest.mark.parametrize('name', AL.CASES)
def test_agent_main_loop_matches_reference_loop(name, cpu_double, monkeypatch):
    """Agent.main_setup / main_loop, hooks, fetch cadence, counters (agent/base.py:160-271)"""
    AL.check_case(name, monkeypatch)
