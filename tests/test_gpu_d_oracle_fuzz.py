"""GPU: a slice of the differential fuzzer (tests/tools/oracle_fuzz.py) in the suite -- 40 randomly drawn scenes / cameras / entry points,
every output bit-equal to the C oracle.  Long runs: `python tests/tools/oracle_fuzz.py 1000` (profiles/r05_oracle_fuzz.txt)."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
import oracle_fuzz as fz  # noqa: E402

pytestmark = pytest.mark.gpu


def test_forty_drawn_cases_are_bit_equal_to_the_oracle(rasterizer):
    entries, degrees = set(), set()
    for seed in range(40):
        c = fz.draw_case(seed)
        diffs = fz.run_case(rasterizer, c)
        assert not diffs, (fz.describe(c), diffs)
        entries.add(c["entry"]); degrees.add(c["deg"])
    assert entries == {"single", "batch", "posed", "host", "pipelined"} and degrees == {-1, 0, 1, 2, 3}      # the slice reaches every arm
