import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    r = d.get("roofline") or {}
    print(f, "G edges/s", round(d["value"] / 1e9, 3), "ms/step", round(d["ms_per_step"], 5), "projected", (d["config"].get("projected_input") or {}).get("precompute_s"),
          {k: (v["ms_per_step"], v["frac"]) for k, v in (r.get("by_kernel") or {}).items()}, "dominant", r.get("dominant"), r.get("frac"))
