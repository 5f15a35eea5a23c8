#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06o; mkdir -p $o
timeout 1500 python -m pytest "tests/test_gpu_entry_points.py::test_bench_emulated_world_line" tests/test_bench_launcher.py -x -q -m gpu 2>&1 | tail -6
for g in 16 32; do
t0=$(date +%s); timeout 600 python bench.py --workload mag240m-sharded --emulate-world 8 --shard-scale 0.08 --fanouts 25,10 --batch 1024 --shard-group $g --steps 256 --no-cpu-baseline --no-live-pmc > $o/sub_g$g.json 2> $o/sub.err; echo "G=$g child wall $(( $(date +%s) - t0 )) s"; tail -2 $o/sub.err
python - $o/sub_g$g.json <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(round(j["value"]/1e9,2), "G", j["route"], round(j["ms_per_step"]*1e3,2)); print({k:v for k,v in j["world1_reference"].items() if k!="measured"})
for t,e in j["emulated"].items():
    if isinstance(e,dict) and e.get("overlapped"):
        pj=e["projection"]
        print(t, "kernel", round(e["kernel_ms_per_step_per_rank"]*1e3,1), "ovl", round(e["overlapped"]["ms_per_rank_step"]*1e3,1), "link", round(pj["link_ms_per_step"]*1e3,1), {k:round(v,2) for k,v in pj.items() if k.startswith("scaling")}, e["kernel_ms_by_group"])
PY
done
timeout 600 python bench.py --workload mag240m-sharded > $o/bench_world1.json 2>/dev/null; python -c "
import json; j=json.loads([l for l in open('$o/bench_world1.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'], j['config']['hop_route'])"
