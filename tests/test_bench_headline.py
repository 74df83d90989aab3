"""bench.py's stdout contract: ONE compact JSON line the driver can keep whole (round 4's 20 KB line was not parsed).
The canned record is round 4's full bench record (profiles/r04/bench_line_full.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "tests", "golden", "bench_full_record_r04.json")


def _headline():
    return bench.headline(json.load(open(CANNED)))


def test_headline_fits_the_byte_budget():
    line = json.dumps(_headline(), separators=(",", ":"))
    assert len(line) < bench.HEADLINE_MAX_BYTES == 4096
    assert "\n" not in line


def test_headline_carries_what_the_driver_reads():
    h = _headline()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_mismatches_vs_oracle", "summary"):
        assert k in h, k
    assert isinstance(h["config"]["workload"], str) and len(h["config"]["workload"]) < 200
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_check", "traffic_source"):
        assert k in h["roofline"], k
    assert abs(h["roofline"]["frac"] - h["roofline"]["achieved"] / h["roofline"]["peak"]) < 1e-6
    assert h["roofline"]["valu"]["frac"] > 0
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread_value"):
        assert k in h["cpu_baseline"], k
    assert h["cpu_baseline"]["kind"] == "port"
    s = h["summary"]
    assert s["planner"]["ms_per_batch"] > 0 and s["env_steps_per_s"]["push"]["ct"] > 0
    assert set(s["rollout_agent_steps_per_s"]) >= {"lockstep", "async", "async_dyn", "lift", "assembly_ik"}


def test_headline_holds_no_prose_blocks():
    h = _headline()

    def strings(x):
        if isinstance(x, str):
            yield x
        elif isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
    assert max(len(t) for t in strings(h)) < 200


def test_an_oversized_record_still_yields_a_short_line():
    rec = json.load(open(CANNED))
    for i in range(400):
        rec[f"rollout_extra_{i}"] = {"agent_steps_per_s": 1.0 + i, "envs_stepping_per_call": 2.0}
    line = json.dumps(bench.headline(rec), separators=(",", ":"))
    assert len(line) < bench.HEADLINE_MAX_BYTES
    assert json.loads(line)["roofline"]["frac"] > 0 and json.loads(line)["cpu_baseline"]["value"] > 0
