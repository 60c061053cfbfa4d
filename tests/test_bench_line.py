"""The line bench.py prints is what the driver records, and the driver keeps 8 KB of stdout: both halves of the metric must be
inside it (round 4's 17 KB line lost its Merkle half)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compact_line_keeps_both_halves_inside_the_drivers_window():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04zz_bench.json")))  # a real 16 KB record of round 4
    assert len(json.dumps(full)) > 12000
    line = b.compact_line(full)
    s = json.dumps(line)
    assert len(s) <= 7000, len(s)
    # the contract's keys survive untouched in meaning
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["vs_baseline"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"] and k in line["merkle"]["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"] and k in line["merkle"]["cpu_baseline"], k
    # the Merkle half -- value, ms, roofline, cpu_baseline -- ends inside the first 4 KB
    assert s.index('"two_roots_in_flight"') < 4096
    assert abs(line["merkle"]["value"] / full["merkle"]["value"] - 1) < 1e-4
    assert abs(line["value"] / full["value"] - 1) < 1e-4
    # no prose
    assert '"note"' not in s
