#!/bin/bash
# round-6 session 2: emulated 8-rank world, bucketed vs peer-mapped feature route, kernel time + overlapped step
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06b; mkdir -p $o
timeout 900 python bench.py --workload mag240m-sharded --emulate-world 8 --shard-route both > $o/bench_emulated_world8.json 2> $o/emu.err
tail -5 $o/emu.err
python - <<PY
import json
j=json.load(open("$o/bench_emulated_world8.json"))
print(j["value"], j["ms_per_step"], j.get("route"))
for t,e in j["emulated"].items():
    if not isinstance(e,dict) or "route" not in e: continue
    print(t, e["route"], "kernel_ms", round(e["kernel_ms_per_step_per_rank"],5), "share", e["sharded_only_kernel_share"], "overl", e["overlapped"], "pulled", round(e["pulled_rows_per_step_mean"]), "link_ms", round(e["projection"]["link_ms_per_step"],5))
    print("   ", e["kernel_ms_by_group"])
    print("   ", {k:v for k,v in e["projection"].items() if k.startswith("whole") or k.startswith("overl")})
PY
timeout 600 python bench.py --workload mag240m-sharded > $o/bench_mag240m-sharded.json 2> $o/sh.err; tail -2 $o/sh.err; head -c 400 $o/bench_mag240m-sharded.json; echo
timeout 600 python bench.py --workload mag240m-sharded --shard-route peer > $o/bench_mag240m-sharded_peer.json 2> $o/shp.err; tail -2 $o/shp.err; head -c 400 $o/bench_mag240m-sharded_peer.json; echo
