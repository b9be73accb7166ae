"""GPU tier: DDPGLearner on the HIP path against the reference goldens (1e-5)."""
import pytest

import ddpg_helpers as DH

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', DH.DDPG_CASES)
def test_ddpg_learner_matches_reference_golden(name):
    DH.run_and_check(name)
