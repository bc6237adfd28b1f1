"""The bench record's final line stays parseable by the driver: < 4 KB, strict JSON, the contract's keys (round 5's
21.8 KB line left BENCH_r05.parsed = null).  Built from the canned full record of round 5's own run (gpurun_out is not shipped, so a
trimmed copy lives in tests/golden/bench_full_record_r05.json) with the arrays inflated and NaNs planted."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_record  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def canned():
    with open(os.path.join(ROOT, "tests", "golden", "bench_full_record_r05.json")) as f:
        return json.load(f)


def strict_loads(line):
    def no_constants(name):
        raise ValueError("non-finite constant " + name)
    return json.loads(line, parse_constant=no_constants)


def test_final_line_is_short_strict_json_with_the_contract_keys():
    rec = canned()
    detail, sec, final = bench_record.lines(rec)
    assert len(final) < 4096 and "\n" not in final
    o = strict_loads(final)
    for k in CONTRACT:
        assert k in o, k
    assert list(o["config"])[0] == "workload" and len(o["config"]) <= 20
    assert all(not isinstance(v, (dict, list)) for v in o["config"].values())
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in o["roofline"], k
    assert all(not isinstance(v, (dict, list)) for v in o["roofline"].values())
    assert abs(o["roofline"]["frac"] - o["roofline"]["achieved"] / o["roofline"]["peak"]) < 1e-5
    assert set(o["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert o["value"] == float(f"{rec['value']:.6g}") and o["steps"] == rec["steps"] and o["warmup"] == rec["warmup"]
    assert dumps_roundtrip(o)


def dumps_roundtrip(o):
    return strict_loads(json.dumps(o, allow_nan=False)) == o


def test_secondary_line_carries_c3_and_c5_with_their_own_roofline_and_cpu_baseline():
    detail, sec, final = bench_record.lines(canned())
    assert sec.startswith("bench_secondary ") and len(sec) < 3600 + len("bench_secondary ")
    o = strict_loads(sec[len("bench_secondary "):])
    for k in ("ransac_c3", "c5"):
        assert o[k]["roofline"]["frac"] > 0 and o[k]["cpu_baseline"]["cores"] == 1 and o[k]["value"] > 0, k
    # both lines together fit the driver's 8 KB stdout tail
    assert len(sec) + len(final) + 2 < 8192
    assert detail.startswith("bench_detail ") and strict_loads(detail[len("bench_detail "):])["metric"]


def test_growth_and_nan_do_not_break_the_final_line():
    rec = canned()
    rec["extras"] = {f"extra_{i}": {"note": "x" * 500, "ms_per_call": float(i)} for i in range(200)}   # a 100 KB record
    rec["timed_region"] = {"ms_steps_gpu": [0.4] * 5000}
    rec["roofline"]["note"] = "y" * 5000
    rec["roofline"]["kernel_ms_avg"] = float("nan")
    rec["config"]["steady_frac"] = float("inf")
    rec["config"]["workload"] = "w" * 1000
    rec["cpu_baseline"]["sample"] = "s" * 1000
    detail, sec, final = bench_record.lines(rec)
    assert len(final) < 4096
    o = strict_loads(final)
    assert o["roofline"]["kernel_ms_avg"] is None and o["config"]["steady_frac"] is None
    assert len(o["config"]["workload"]) <= 120 and len(o["cpu_baseline"]["sample"]) <= 96
    assert not any(isinstance(v, float) and not math.isfinite(v) for v in o["roofline"].values())


def test_the_c5_workload_record_goes_through_the_same_path():
    rec = canned()["c5"]
    rec["c5"] = dict(rec)
    detail, sec, final = bench_record.lines(rec)
    o = strict_loads(final)
    assert len(final) < 4096 and o["unit"] == "image-pairs/s" and list(o["config"])[0] == "workload" and o["roofline"]["frac"] > 0
