"""GPU tier (-m gpu): learn / publish sequences on the HIP path against the reference's own
recorded sequences (tests/golden/ppo_sequences.json) -- with hipGraph replay (the default: the
adapted beta / clip_epsilon reach the captured step through the device control block) and
eagerly."""
import pytest

import sequence_cases as SC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', SC.NAMES)
def test_learn_publish_sequence_matches_reference_graph(name):
    L = SC.run_sequence(name)
    assert L.use_graph and len(L._graphs) >= 1


@pytest.mark.parametrize('name', [n for n in SC.NAMES if not n.startswith('cfg5')])
def test_learn_publish_sequence_matches_reference_eager(name):
    SC.run_sequence(name, session_overrides={'use_hip_graph': False, 'lazy_stats': False})
