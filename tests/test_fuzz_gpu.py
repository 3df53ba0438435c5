"""A handful of seeds of every fuzzer in the suite (round 3: three of the fuzzers' first scenes hit edge cases the hand-written
tests had missed -- a hung rt_flags_kernel, a walk into unmapped memory, a host division by zero).  The generators live in tools/
(wider sweeps: `python tools/fuzz_*.py first last`); these seeds are ones no earlier sweep has run."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_fuzz_sweep_generator_1_against_the_oracle():
    import fuzz_sweep
    assert fuzz_sweep.run(9001, 9005, 1) == 0


def test_fuzz_sweep_generator_2_against_the_oracle():
    import fuzz_sweep
    assert fuzz_sweep.run(9001, 9005, 2) == 0


def test_fuzz_knobs_image_independent_of_every_execution_knob():
    import fuzz_knobs
    assert fuzz_knobs.run(9001, 9005) == 0


def test_fuzz_sequence_random_frame_sequences_on_one_scene_handle():
    import fuzz_sequence
    assert fuzz_sequence.run(9001, 9005) == 0
