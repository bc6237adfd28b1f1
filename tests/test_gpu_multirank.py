"""N > 1 where a GPU exists: bench.py's C5 workload with TWO ranks (gloo, both on cuda:0 -- a rehearsal of the sharding, the all_gather
of the records and the grouped send / recv of the match lists, not a scaling run) against the same batch on one rank: gathered records
and match lists byte for byte.  The ranks are fresh child processes started by bench.py's own launcher (a process that never touched the
GPU) under torch.distributed.run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, gpus, estimator="ransac"):
    out = os.path.join(str(tmp_path), f"rec{gpus}_{estimator}.npz")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", "c5", "--c5-pairs", "16", "--c5-distinct", "16", "--n", "2048", "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline", "--dump-records", out, "--estimator", estimator]
    if gpus > 1:
        cmd += ["--share-gpu", "--backend", "gloo"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    final = lines[-1]                                   # what the driver parses: the LAST line, short, strict JSON
    assert final.startswith("{") and len(final) < 4096, len(final)
    head = json.loads(final)
    assert head["n_gpus"] == gpus and head["config"]["workload"].startswith("C5") and head["roofline"]["frac"] > 0
    if gpus > 1:   # the record itself proves the collective library saw every rank
        assert head["rccl_ranks_seen"] == {"world_size": gpus, "all_reduce_of_ones": gpus, "backend": "gloo"}
    detail = [ln for ln in lines if ln.startswith("bench_detail ")][-1]
    return json.loads(detail[len("bench_detail "):]), np.load(out)


def test_two_ranks_give_the_one_rank_records_and_match_lists(tmp_path):
    one_line, one = _run(tmp_path, 1)
    two_line, two = _run(tmp_path, 2)
    assert one_line["n_gpus"] == 1 and two_line["n_gpus"] == 2 and two_line["config"]["backend"] == "gloo"
    assert two_line["config"]["pairs_total"] == 16 and two_line["config"]["pairs_this_rank"] == 8
    assert one["records"].tobytes() == two["records"].tobytes()
    assert one["matches"].shape == two["matches"].shape and one["matches"].tobytes() == two["matches"].tobytes()
    assert len(one["matches"]) > 16 * 100


@pytest.mark.parametrize("estimator", ["usac_prosac", "arrsac"])
def test_two_ranks_with_the_sequential_estimators(tmp_path, estimator):
    """the sharded C5 workload with the harness' default estimator (USAC, PROSAC by matching cost) and with estimateEssentialMat's
    default (ARRSAC) -- every rank runs its share through the launch hub (fibers on its share of the host's cores) -- gathers the records and
    match lists the one-rank run produces, byte for byte."""
    one_line, one = _run(tmp_path, 1, estimator)
    two_line, two = _run(tmp_path, 2, estimator)
    assert one_line["config"]["estimator"] == estimator and two_line["config"]["estimator"] == estimator
    ht = two_line["config"]["host_threads"]
    assert ht["local_world_size"] == 2 and 2 <= ht["hub_workers_per_cohort"] <= 16
    assert two_line["n_gpus"] == 2 and two_line["config"]["pairs_this_rank"] == 8
    assert one["records"].tobytes() == two["records"].tobytes()
    assert one["matches"].tobytes() == two["matches"].tobytes()


@pytest.mark.parametrize("workload", ["c2", "c5"])
def test_every_collective_of_the_n_gpu_path_runs_through_rccl_with_one_rank(tmp_path, workload):
    """No box of the pool has two cards, so RCCL never sees N > 1 here -- but every collective CALL of the N > 1 path can run for real with a
    world of one: `--force-dist` initialises the nccl process group on cuda:0 and takes every `N > 1` branch (the all_reduce of ones, the
    asynchronous all_gather of the match counts per step, the staged all_gather of the records through pinned + device buffers, the
    max-over-ranks reduction, the barriers).  The record must say so, and the C5 records must equal the plain one-rank run's."""
    out = os.path.join(str(tmp_path), f"force_{workload}.npz")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--settle-steps", "0",
            "--steady-steps", "0", "--after-idle-launches", "0"]
    if workload == "c5":
        base += ["--workload", "c5", "--c5-pairs", "16", "--c5-distinct", "16", "--n", "2048", "--dump-records", out]
    else:
        base += ["--no-extras", "--pairs-per-gpu", "8", "--n", "2048"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(base + ["--force-dist"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    head = json.loads(r.stdout.splitlines()[-1])
    assert head["rccl_ranks_seen"] == {"world_size": 1, "all_reduce_of_ones": 1, "backend": "nccl"}, head.get("rccl_ranks_seen")
    assert head["n_gpus"] == 1 and head["value"] > 0
    if workload == "c5":
        forced = np.load(out)
        plain_out = os.path.join(str(tmp_path), "plain.npz")
        r2 = subprocess.run([a if a != out else plain_out for a in base], env=env, capture_output=True, text=True, timeout=600)
        assert r2.returncode == 0, r2.stderr[-3000:]
        plain = np.load(plain_out)
        assert forced["records"].tobytes() == plain["records"].tobytes() and forced["matches"].tobytes() == plain["matches"].tobytes()
