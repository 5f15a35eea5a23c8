#!/bin/bash
# round-6 session 9: which feature route wins at what hot-row fraction; the sampler entry at the job's call size
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06i; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_dist_peer.py tests/test_gpu_entry_points.py -x -q -m gpu 2>&1 | tail -3
for hf in 0 0.01 0.05; do
  timeout 900 python bench.py --workload mag240m-sharded --emulate-world 8 --shard-route both --shard-hot-frac $hf --shard-scale 0.12 > $o/emu_hot_$hf.json 2> $o/emu.err || tail -3 $o/emu.err
  python - $o/emu_hot_$hf.json $hf <<PY
import json,sys
j=json.load(open(sys.argv[1]))
for t,e in j["emulated"].items():
    if not isinstance(e,dict) or "route" not in e or not e.get("overlapped"): continue
    pj=e["projection"]
    print("hot", sys.argv[2], t, "kernel us", round(e["kernel_ms_per_step_per_rank"]*1e3,1), "overlapped us", round(e["overlapped"]["ms_per_rank_step"]*1e3,1), "rows", round(e["pulled_rows_per_step_mean"]), "link us", round(pj["link_ms_per_step"]*1e3,1), "G hidden", round(pj["whole_node_edges_per_s_overlapped_links_hidden"]/1e9,1), "G not hidden", round(pj["whole_node_edges_per_s_overlapped_links_not_hidden"]/1e9,1))
PY
done
timeout 900 python bench.py --entry sampler --no-cpu-baseline > $o/bench_entry_sampler.json 2> $o/es.err; tail -2 $o/es.err
python - <<PY
import json
j=json.loads([l for l in open("$o/bench_entry_sampler.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["config"]["workload"][:200]); print({k:v for k,v in j.get("roofline",{}).items() if k in ("kernel","frac","achieved")})
PY
