"""A short leg of the randomised differential soak (tools/fuzz_parity.py) inside the GPU suite: random points of the
configuration space, HIP library vs oracle, free-running with auto-reset, every byte equal at every step."""
import importlib.util
import os
import random

import pytest

import oracle_env

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_bit_exact(seed):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(REPO, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    oracle_env.set_math_mode(1)
    try:
        rng = random.Random(seed)
        for c in range(40):
            E, N, steps, probs, kw = fz.draw_case(rng)
            assert fz.run_case(E, N, steps, probs, kw) is None, (c, E, N, steps, kw)
    finally:
        oracle_env.set_math_mode(0)
