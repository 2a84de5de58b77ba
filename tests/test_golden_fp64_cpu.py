"""CPU: the float64 yardstick of the 1e-4 gradient bar is pinned to the REFERENCE.

tests/golden/<case>_fp64.npz were written by tests/golden/make_golden_fp64.py, which imports /root/reference/model/network.py,
casts it to float64 and runs it through its dense tuple input form.  Here the oracle (oracle/dense_ref.py) in float64 -- driven
through the recording machinery of tests/discrete.py that every large parity test relies on -- must reproduce the reference's
gradients, pre-activations, readout operands and winners."""
import pytest

import discrete
from util import CASES


@pytest.mark.parametrize('name', CASES)
def test_fp64_oracle_and_decision_recording_match_the_reference_fp64_fixture(name):
    discrete.check_machinery_against_reference_fp64(name)
