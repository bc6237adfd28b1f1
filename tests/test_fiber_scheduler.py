"""The run scheduler of the launch hub (csrc/batch_hub.h) without a GPU: fibers on worker threads handing over to a hub through futex
words.  mlpl_debug_fiber_selftest drives n fibers through `rounds` hand-overs and checks that every fiber is released exactly once per
round; here with one worker for all fibers (every wait is a context switch on one thread), with more workers than fibers, and with far
more fibers than this machine has cores."""
import pytest

from matchinglib_poselib_amd import _lib


@pytest.mark.parametrize("n,workers,rounds", [(1, 1, 1), (7, 1, 50), (128, 16, 40), (3, 8, 200), (600, 5, 12), (64, 64, 100)])
def test_every_fiber_is_released_once_per_round(n, workers, rounds):
    lib = _lib.load_library()
    assert lib.mlpl_debug_fiber_selftest(n, workers, rounds) == n * rounds


def test_bad_arguments_are_refused():
    lib = _lib.load_library()
    assert lib.mlpl_debug_fiber_selftest(0, 1, 1) == -1 and lib.mlpl_debug_fiber_selftest(1, 0, 1) == -1


def test_repeated_runs_reuse_nothing_stale():
    lib = _lib.load_library()
    for _ in range(20):
        assert lib.mlpl_debug_fiber_selftest(33, 4, 5) == 165
