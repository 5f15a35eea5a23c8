#!/bin/bash
# round-5 session 19 (GPU box): one bench line per workload on the tree as committed, the products profile round
# (kernel stats, overlap, PMC passes, SQ counters of the two dominant kernels), the emulated 8-rank world
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05r; mkdir -p $o
b() { local name=$1; shift; timeout 900 python bench.py "$@" 2> $o/bench_$name.err | grep '^{' | tail -1 > $o/bench_$name.json; }
b products_driver_flags --steps 20 --warmup 5
b mag-shard --workload mag-shard --no-live-pmc --steps 20 --warmup 5
b mag240m-sharded --workload mag240m-sharded --steps 20 --warmup 5
b emulated_world8 --workload mag240m-sharded --emulate-world 8 --no-cpu-baseline --steps 256
b cora --workload cora --no-live-pmc --steps 20 --warmup 5
b gat-lp --workload gat-lp --steps 20 --warmup 5
b gat-lp_train --workload gat-lp --train --steps 20 --warmup 5
b train --train --steps 20 --warmup 5
b lp_train --train --train-task lp --steps 64 --warmup 8
b entry_sampler --entry sampler --steps 20 --warmup 5
b entry_inferencer --entry inferencer --steps 20 --warmup 5
GIGL_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 20 --warmup 5 2> $o/bench_n2_shared_gpu.err | grep '^{' | tail -1 > $o/bench_n2_shared_gpu.json
scripts/prof_round.sh r05r products "gather_mean|linear_fused2" > $o/prof_round.log 2>&1
python - <<P
import json, glob
for f in sorted(glob.glob("$o/bench_*.json")):
    try:
        d = json.load(open(f))
        r = d.get("roofline") or {}
        print(f.split("/")[-1], d["metric"][:40], "value", "%.4g" % d["value"], "ms/step", round(d["ms_per_step"], 5), "roofline", r.get("kernel"), r.get("frac"))
    except Exception as e:
        print(f.split("/")[-1], "no line", e)
P
