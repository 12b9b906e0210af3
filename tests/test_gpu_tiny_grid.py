"""The suite's small segments on a grid sized for ONE compute unit (PINOT_GPU_TEST_CUS=1, read when a segment is opened): every
persistent kernel's tile loop then runs many times per wavefront on a 100 000-doc segment -- second tiles, carried accumulators, the
records of waves that end early.  DESIGN.md 4.3f: `group_private_kernel`'s direct-table forms were wrong from a wave's second tile on and
passed every test below 4.5 M docs on the full 256-CU grid for two rounds.  The WHOLE GPU suite passes under the switch
(`PINOT_GPU_TEST_CUS=1 python -m pytest tests -m gpu`, profiles/r4/gpu_suite_one_cu_grid.txt); the driver's plain run gets the whole fuzz
(tests/test_gpu_fuzz.py, every seed) under this grid as well as the full one, the samples below, and tests/test_gpu_kernel_coverage.py --
every kernel of the library dispatched under this grid and on a 12.3 M-doc segment."""
import pytest

import test_gpu_fuzz as F
import test_gpu_group_map as G
import test_gpu_index_and as I
import test_gpu_typed as T

pytestmark = pytest.mark.gpu


@pytest.fixture()
def one_cu(monkeypatch):
    monkeypatch.setenv("PINOT_GPU_TEST_CUS", "1")


@pytest.mark.parametrize("seed", list(range(24)))          # (round 5: every seed of the fuzz under both grids, not a sample of four)
def test_random_segments_and_queries(engine, one_cu, seed):
    F.test_random_segments_and_queries(engine, seed)


@pytest.mark.parametrize("seed", list(range(16)))
def test_null_vectors_and_wide_group_bys(engine, one_cu, seed):
    F.test_random_null_vectors_null_handling_and_wide_group_bys(engine, seed)


def test_partitioned_and_two_level_group_by(engine, one_cu):
    G.test_partitioned_path_on_a_segment_large_enough_to_take_it(engine)
    G.test_two_level_partitioning_above_two_million_keys(engine, (3000, 2500), 4_300_000)


def test_postings_and_typed_columns(engine, one_cu):
    I.test_and_of_postings_in_every_container_kind(engine, 400_009, False)
    T.test_raw_typed_range_filters_and_fallbacks(engine)
