import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print("streams", d["config"]["streams"], "value %.3f G" % (d["value"] / 1e9), "ms/step %.5f" % d["ms_per_step"], r["kernel"],
      "frac", r["frac"], "alone", r.get("frac_alone"), r["overlapped_ms_per_step"])
