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


def test_round6_line_with_latency_curve_and_config2_readings_fits_7_kb():
    """the line carries the whole latency curve (11 sizes) and SURVEY 8(d) config 2's other readings and
    still fits 7 KB; both rooflines keep their `traffic` and say where it comes from."""
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06j_bench_full.json")))  # the full record of the round's final visit
    line = b.compact_line(full)
    s = json.dumps(line)
    assert len(s) <= 7168, len(s)
    lc = line["latency_curve"]
    assert lc["n"] == [1, 64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536] and len(lc["ms"]) == 11
    assert lc["sigs_per_s_never_drops_below"] >= 0.9 and lc["check"]["statuses_match_construction"]
    assert set(lc["thresholds"]) == {"rows_up_to", "lane_groups_up_to", "two_lanes_up_to", "measured_on_this_device"}
    for shape in ("one_call_k65536", "n64_k1024"):
        for sem in ("reference_semantics", "validated_key_registry"):
            assert line["config2_readings"][shape][sem]["ms"] > 0
    assert "first_call" in line["block"]["scalar_call"]
    assert s.index('"latency_curve"') < 5000 and '"note"' not in s
