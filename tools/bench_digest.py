#!/usr/bin/env python3
"""One-screen digest of a bench.py JSON line (development aid for GPU visits)."""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]).read().strip().splitlines() if ln.startswith("{")][-1])
r = d.get("roofline", {})
print("step %.2f ms  %.0f %s  kernel %s %.2f ms  check %s" % (d["ms_per_step"], d["value"], d["unit"], r.get("kernel"), r.get("avg_launch_ms") or 0, d.get("check")))
if "stage_ms" in r:
    print("  stages", {k: round(v, 2) for k, v in r["stage_ms"].items()}, "T mult/s", {k: round(v, 1) for k, v in r["valu_int"]["achieved"].items()})
if r.get("pairing_parts_ms"):
    print("  pairing parts", {k: round(v, 2) for k, v in r["pairing_parts_ms"].items()})
for key in ("aggregates_k2048", "merkle", "epoch", "slots", "strong_2p20", "half_round_32768", "merkle_strong", "merkle_sharded_emulated"):
    if key in d:
        e = d[key]
        print("  %s: %.3f ms/step  %.4g %s  check %s" % (key, e.get("ms_per_step", 0), e.get("value", 0), e.get("unit"), e.get("check")))
        if e.get("phases"):
            print("      phases", {k: round(v, 3) for k, v in e["phases"].items() if k.endswith("_ms") or k.endswith("rank")})
if "block" in d:
    print("  block", round(d["block"]["reference_semantics"]["block_verify_ms"], 2), round(d["block"]["validated_key_registry"]["block_verify_ms"], 2),
          {k: round(v, 2) for k, v in d["block"].get("scalar_call", {}).items() if isinstance(v, (int, float))})
if "msm" in d:
    print("  msm", {k: (round(v["ms"], 2), round(v["points_per_s"])) for k, v in d["msm"].items() if k.startswith("g1_") and isinstance(v, dict) and "ms" in v},
          d["msm"].get("check"))
if "box_selfcheck" in d:
    print("  box slowdown", round(d["box_selfcheck"]["large_code_slowdown"], 2), d["box_selfcheck"]["pairing_kernels"])
if "merkle" in d and d["merkle"].get("h2d_inclusive"):
    print("  merkle incl. H2D", {k: round(v, 2) for k, v in d["merkle"]["h2d_inclusive"].items() if k.endswith("_ms")})
if "merkle" in d and d["merkle"].get("two_roots_in_flight"):
    t = d["merkle"]["two_roots_in_flight"]
    print("  merkle, two roots in flight: %.3f ms per root  %.4g leaves/s  %s" % (t["ms_per_root"], t["leaves_per_s"], t["roots_equal_the_one_stream_root"]))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("  cpu", round(c["value"]), c["unit"], "cores", c.get("cores"), c.get("cores_effective"))
