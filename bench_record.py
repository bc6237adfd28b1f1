"""The lines bench.py prints, built from its full measurement record (no GPU, no torch: tests/test_bench_record.py runs it on a canned record).

The driver keeps the last ~8 KB of stdout and parses the LAST line: round 5's 21.8 KB line was cut and `BENCH_r05.parsed` came out null.
So the record is split:
  * `bench_detail.json` next to bench.py and ONE earlier stdout line `bench_detail {...}`: everything (extras, per-step arrays, notes);
  * one earlier stdout line `bench_secondary {...}` (<= 3.5 KB): BASELINE's second metric half (C3 RANSAC hyp/s) and config 5, each with its
    own `roofline` and `cpu_baseline` -- it sits inside the driver's tail together with the final line;
  * the FINAL line (< 4 KB, strict JSON, scalars and short strings only): the bench contract's fields, `config` (<= 20 scalar keys,
    `workload` first), `roofline` (10 scalar keys), `cpu_baseline`.
"""
import json
import math

FINAL_LINE_MAX = 4096
SECONDARY_LINE_MAX = 3584

ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_avg", "launches_timed", "mfma_busy_frac")
CONFIG_KEYS = ("workload", "pairs_per_gpu", "value_single_pair", "ms_single_pair", "value_8_pairs_per_launch", "steady_pairs_per_s", "steady_frac",
               "clock_GHz_steady_median", "clock_GHz_stamped_median", "ransac_c3_hyp_per_s", "ransac_c3_count_frac_fp32_peak",
               "ransac_c3_cpu_hyp_per_s_1_core", "c5_image_pairs_per_s", "c5_ms_per_step", "c5_estimator", "c5_cpu_image_pairs_per_s_1_core",
               "c5_usac_default_refine_ms_per_512", "settle_steps", "solver_polish", "verified")


def _num(x, sig=6):
    """A JSON-safe scalar: floats to `sig` significant digits, non-finite floats to None (strict JSON has no NaN / Infinity)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        f = float(x)
    except (TypeError, ValueError):
        return str(x)[:96]
    if not math.isfinite(f):
        return None
    if f == 0:
        return 0.0
    return float(f"{f:.{sig}g}")


def _short(s, n=96):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _pick(d, keys, sig=6, strlen=96):
    return {k: _short(_num(d.get(k), sig), strlen) for k in keys if d is not None and k in d}


def _cpu(c, strlen=96):
    if not c:
        return None
    return {"value": _num(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"), "sample": _short(c.get("sample"), strlen)}


def headline(rec):
    """The final stdout line's object."""
    out = {k: _num(rec.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                         "vs_baseline", "dtype", "data")}
    out["dtype"] = _short(out["dtype"], 48)
    cfg = rec.get("config") or {}
    keys = [k for k in CONFIG_KEYS if k in cfg]
    if len(keys) < 3:   # another workload's record (--workload c5): `workload` and its first scalars
        keys = [k for k, v in cfg.items() if not isinstance(v, (dict, list, tuple))][:20]
    out["config"] = {k: _short(_num(cfg.get(k)), 120 if k == "workload" else 96) for k in keys}
    assert len(out["config"]) <= 20
    rl = rec.get("roofline") or {}
    out["roofline"] = _pick(rl, ROOFLINE_KEYS, strlen=48)
    if "kernel_ms_avg" not in rl and "kernel_ms_per_step" in rl:
        out["roofline"]["kernel_ms_per_step"] = _num(rl["kernel_ms_per_step"])
    out["cpu_baseline"] = _cpu(rec.get("cpu_baseline"))
    if "rccl_ranks_seen" in rec:
        out["rccl_ranks_seen"] = rec["rccl_ranks_seen"]
    return out


def _sub(o, cfg_keys):
    """A secondary measurement (C3 / C5) with its own roofline and cpu_baseline, scalars only."""
    if not o:
        return None
    r = {k: _short(_num(o.get(k)), 100) for k in ("metric", "value", "unit", "ms_per_step", "ms_per_call", "n_gpus", "steps", "mode", "scaling", "dtype")
         if k in o}
    if isinstance(o.get("config"), dict):
        r["config"] = _pick(o["config"], cfg_keys, strlen=64)
    if isinstance(o.get("roofline"), dict):
        r["roofline"] = _pick(o["roofline"], ROOFLINE_KEYS[:7] + ("kernel_ms", "kernel_ms_per_step", "launches_timed", "solver_frac_of_fp64_vector_peak",
                                                                 "hamming_kernel_frac_of_fp4_peak"), strlen=48)
    r["cpu_baseline"] = _cpu(o.get("cpu_baseline"), 80)
    return r


def secondary(rec):
    """The `bench_secondary` line's object: C3 (RANSAC hyp/s) and C5 (image pairs/s), the USAC forms of C5 as scalars."""
    out = {"ransac_c3": _sub(rec.get("ransac"), ()),
           "c5": _sub(rec.get("c5"), ("estimator", "pairs_total", "pairs_this_rank", "world_size", "mean_matches", "mean_inliers", "iterations_rank0"))}
    if isinstance((rec.get("c5") or {}).get("kernel_ms_per_step_rank0"), dict):
        out["c5"]["kernel_ms_per_step"] = {_short(k, 40): _num(v, 4) for k, v in rec["c5"]["kernel_ms_per_step_rank0"].items()}
    ex = rec.get("extras") or {}
    usac = {}
    for name in ("c5_usac_batch_uniform", "c5_usac_batch_prosac", "c5_usac_batch_default_refinement_prosac"):
        e = ex.get(name)
        if isinstance(e, dict):
            usac[name.replace("c5_usac_batch_", "")] = {"ms_per_512_pairs": _num(e.get("ms_per_step"), 4), "image_pairs_per_s": _num(e.get("value"), 4),
                                                       **{k: _num(v, 4) for k, v in (e.get("hub_last_internal_call") or {}).items()}}
    if usac:
        out["c5_usac"] = usac
    single = {k: _num(ex[k].get("ms_per_call"), 4) for k in ("hamming_c2_single_pair", "usac_uniform", "usac_prosac", "arrsac_default_method",
                                                            "l2_c4_auto", "l2_c4_rootsift_auto") if isinstance(ex.get(k), dict)}
    if single:
        out["ms_per_call"] = single
    return out


def dumps(o):
    return json.dumps(o, allow_nan=False, separators=(",", ":"))


def lines(rec):
    """(detail line, secondary line, final line) -- strict JSON each; the final one shorter than FINAL_LINE_MAX, the secondary one than
    SECONDARY_LINE_MAX (keys are shed from the end rather than the line lost)."""
    h = headline(rec)
    final = dumps(h)
    while len(final) >= FINAL_LINE_MAX and len(h["config"]) > 1:   # never lose the line: shed the last `config` keys instead
        h["config"].popitem()
        final = dumps(h)
    so = secondary(rec)
    sec = dumps(so)
    for k in ("ms_per_call", "c5_usac", "c5", "ransac_c3"):
        if len(sec) < SECONDARY_LINE_MAX:
            break
        so.pop(k, None)
        sec = dumps(so)
    assert len(final) < FINAL_LINE_MAX, len(final)
    json.loads(final), json.loads(sec)
    detail = json.dumps(_sanitize(rec), allow_nan=False)
    return "bench_detail " + detail, "bench_secondary " + sec, final


def _sanitize(o):
    if isinstance(o, dict):
        return {str(k): _sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_sanitize(v) for v in o]
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, (bool, int, str)) or o is None:
        return o
    try:
        f = float(o)
        return f if math.isfinite(f) else None
    except (TypeError, ValueError):
        return str(o)


def emit(rec, detail_path=None):
    """Write bench_detail.json, print the three lines (the final one LAST)."""
    detail, sec, final = lines(rec)
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                f.write(detail[len("bench_detail "):] + "\n")
        except OSError:
            pass
    print(detail, flush=True)
    print(sec, flush=True)
    print(final, flush=True)
