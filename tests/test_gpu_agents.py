"""GPU tier (-m gpu): PPOAgent.act / act_batch and DDPGAgent.act / act_batch on the HIP path
(C ABI -> gfx950 kernels) against the fixtures recorded from the REFERENCE's own
surreal.agent.PPOAgent / DDPGAgent (oracle/gen_golden_agents.py): pd, actions, onetime_infos
at 1e-5."""
import pytest

import agent_cases as AC
import agent_loop_cases as AL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', AC.PPO_CASES)
def test_ppo_agent_matches_reference_agent_hip(name):
    AC.check_ppo_case(name)


@pytest.mark.parametrize('name', AC.DDPG_CASES)
def test_ddpg_agent_matches_reference_agent_hip(name):
    AC.check_ddpg_case(name)


@pytest.mark.parametrize('name', AL.CASES)
def test_agent_main_loop_matches_reference_loop_hip(name, monkeypatch):
    """the reference's own Agent.main_setup / main_loop recording (hook order, fetch cadence, counters,
    per-step actions and distributions, experience windows) replayed with the policy on the HIP path"""
    AL.check_case(name, monkeypatch)
