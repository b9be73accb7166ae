"""CPU tier: learn / publish sequences (host logic: reward filter state, exp_counter, beta /
clip_epsilon adaptation, reference-policy refresh) through the CPU kernel double against the
reference's own recorded sequences; tests/test_gpu_sequences.py runs them on the HIP path."""
import pytest

import sequence_cases as SC


@pytest.mark.parametrize('name', [n for n in SC.NAMES if not n.startswith('cfg5')])
def test_learn_publish_sequence_matches_reference(name, cpu_double):
    SC.run_sequence(name)
