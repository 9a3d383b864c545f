#!/usr/bin/env python3
"""Short view of a bench.py JSON line (file given, or the last line of a log)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f %s  ms_per_step %.2f  n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
print("digits", d["config"].get("digits"), "certificate", d["config"].get("certificate"))
s = d["roofline_secondary"]
print("step", {k: round(v, 2) for k, v in s["step_breakdown_ms"].items()})
print("roofline frac %.4f kernel_ms %.2f traffic %s" % (d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"].get("traffic")))
for k, v in s.items():
    if not isinstance(v, dict) or k == "step_breakdown_ms":
        continue
    r = v.get("roofline") or {}
    print("  %-28s" % k, {kk: (round(v[kk], 3) if isinstance(v[kk], float) else v[kk]) for kk in
                         ("value", "ms_per_step", "slices", "slices_cut", "markers_per_s", "ms_per_call", "frac", "ms") if kk in v},
          ("roofline.frac %.3f ms %.2f" % (r["frac"], r.get("kernel_ms", 0.0))) if r else "")
print("mmt_build_s", d.get("mmt_build_s"), "parity", d.get("parity"))
print("cpu", {k: (d.get("cpu_baseline") or {}).get(k) for k in ("value", "cores", "kind")})
