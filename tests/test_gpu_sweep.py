"""A slice of the randomised parity sweep (tools/parity_sweep.py): small seeded cases with varying sizes,
discretisation, thresholds, masks, force_up, query limits and a few non-finite points; every stage against
the oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_randomised_cases_match_the_oracle(gpu_ctx):
    import parity_sweep
    outcomes = []
    for k in range(24):
        outcomes.append(parity_sweep.one_case(gpu_ctx, 50000 + 7 * (k * 25)))  # incl. seeds 50294's neighbours
    outcomes.append(parity_sweep.one_case(gpu_ctx, 50294))  # tiny raw early-drop sums
    outcomes.append(parity_sweep.one_case(gpu_ctx, 51113))
    assert sum(o.startswith("ok") for o in outcomes) >= 20, outcomes
