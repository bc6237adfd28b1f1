"""CPU: the host-thread budget of a rank at local_world_size = 8 (VERDICT r5 #7a) -- the sequential estimators run as fibers on
`hub_workers` threads per cohort, `lanes` cohorts side by side; eight ranks must not put more runnable threads on the node than twice its
cores, and a rank never falls below one lane with two workers."""
import pytest

from matchinglib_poselib_amd import batch


@pytest.mark.parametrize("estimator", ["usac", "usac_prosac", "usac_default_refine", "arrsac"])
@pytest.mark.parametrize("cpus", [8, 16, 32, 64, 96, 128, 192, 256, 384])
@pytest.mark.parametrize("local_world", [1, 2, 4, 8])
def test_budget_fits_the_node(estimator, cpus, local_world):
    b = batch.host_thread_budget(cpus, local_world, estimator)
    lanes = b["hub_lanes"] or batch.kHubLanesDefault[estimator]
    assert 1 <= lanes <= batch.kHubLanesDefault[estimator] and 2 <= b["hub_workers"] <= 16
    assert b["threads"] == lanes * b["hub_workers"] + lanes + 1
    assert b["cpus_per_rank"] == max(1, cpus // local_world)
    # the node as a whole: every rank's runnable threads within twice the cores (a floor of one lane x two workers per rank aside)
    assert local_world * b["threads"] <= max(batch.kThreadOversubscription * cpus, local_world * 4), (b, cpus, local_world)
    # and a rank with a large share is not throttled below what was measured to matter (8 workers cost 4-7 %, 16 = the library default)
    if cpus // local_world >= 56:
        assert b["hub_lanes"] == 0 and b["hub_workers"] == 16


def test_ransac_needs_no_hub():
    b = batch.host_thread_budget(256, 8, "ransac")
    assert b["hub_workers"] == 0 and b["batch_lanes"] == 2 and b["threads"] == 3


def test_eight_ranks_on_the_bench_box():
    """The GPU boxes of this pool show 256 cores: 32 per rank at --gpus 8."""
    for est, lanes in (("usac", 6), ("arrsac", 4)):
        b = batch.host_thread_budget(256, 8, est)
        assert b["hub_lanes"] == 0 and 8 <= b["hub_workers"] <= 16 and 8 * b["threads"] <= 512, b
