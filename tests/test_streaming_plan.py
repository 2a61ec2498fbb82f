"""Slice planning of the streamed NumPy path (ekm_hip._streamed.plan_slices) -- host logic, no GPU."""
import pytest

from ekm_hip._streamed import leading_axis_bounds, plan_slices

MiB = 1 << 20


def _in_flight(rows, row_bytes, lanes, nslices):
    per = max(hi - lo for lo, hi in leading_axis_bounds(rows, nslices))
    return lanes * per * row_bytes


@pytest.mark.parametrize("rows,row_bytes,budget", [
    (137, 9 * 25 * MiB, 230_000 * MiB),   # config 5 from NumPy: 32 GB working set, fits: 8 lanes
    (137, 9 * 25 * MiB, 8_000 * MiB),     # tight: fewer rows per slice
    (137, 9 * 25 * MiB, 500 * MiB),       # only two single-row slices fit: double buffering
    (64, 12 * MiB, 96 * MiB),
    (721, 4 * 1440 * 8, 64 * MiB),        # config 2 (small rows)
    (3, 100 * MiB, 10_000 * MiB),         # fewer rows than lanes
    (1, 100 * MiB, 10_000 * MiB),
])
def test_plan_fits_the_budget_and_keeps_at_least_two_in_flight(rows, row_bytes, budget):
    lanes, nslices = plan_slices(rows, row_bytes, budget)
    assert 1 <= lanes <= 8 and lanes <= rows and nslices >= lanes
    assert _in_flight(rows, row_bytes, lanes, nslices) <= budget
    if rows >= 2 and rows * row_bytes >= 32 * MiB:
        assert lanes >= 2  # never less than double buffering for anything worth cutting


def test_plan_prefers_many_lanes_when_memory_allows_and_degrades_to_double_buffering():
    assert plan_slices(137, 225 * MiB, 230_000 * MiB) == (8, 120)  # fits: 8 lanes, ~256-MiB slices (short ramp)
    lanes, nslices = plan_slices(137, 225 * MiB, 500 * MiB)
    assert lanes == 2 and nslices == 137  # one row per slice, two in flight
    assert plan_slices(137, 225 * MiB, 400 * MiB) is None  # two rows do not fit: the caller raises


def test_plan_does_not_cut_finer_than_needed():
    lanes, nslices = plan_slices(64, 12 * MiB, 10_000 * MiB)
    assert lanes == 8 and nslices == 8  # everything fits: one slice per lane
    assert plan_slices(0, 1, 1) == (1, 1)
    assert plan_slices(16, 64 << 10, 10_000 * MiB) == (1, 1)  # 1 MiB in total: one slice, one launch
    assert plan_slices(3, 100 * MiB, 10_000 * MiB) == (3, 3)
