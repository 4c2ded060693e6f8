"""GPU: a slice of the differential fuzzer (tests/tools/oracle_fuzz.py) in the suite -- the first drawn scenes / cameras / entry points,
at least 40 and until every entry point, every SH degree and a poisoned scene have come up; every output bit-equal to the C oracle.
Long runs: `python tests/tools/oracle_fuzz.py 4000` (profiles/r05_oracle_fuzz.txt)."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
import oracle_fuzz as fz  # noqa: E402

pytestmark = pytest.mark.gpu
ARMS = {"single", "batch", "posed", "host", "pipelined"}


def test_drawn_cases_are_bit_equal_to_the_oracle(rasterizer):
    entries, degrees, poisoned, seed = set(), set(), 0, 0
    while seed < 40 or not (entries == ARMS and degrees == {-1, 0, 1, 2, 3} and poisoned):
        assert seed < 150, (entries, degrees, poisoned)       # the draw reaches every arm long before
        c = fz.draw_case(seed)
        diffs = fz.run_case(rasterizer, c)
        assert not diffs, (fz.describe(c), diffs)
        entries.add(c["entry"]); degrees.add(c["deg"]); poisoned += c["poisoned"]
        seed += 1
