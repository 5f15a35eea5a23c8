#!/bin/bash
# round-5 session 22 (GPU box): whole -m gpu suite on the tree as it stands; the lines session 19 missed
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05u; mkdir -p $o
timeout 3300 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1
tail -4 $o/pytest_gpu.log
b() { local name=$1; shift; timeout 900 python bench.py "$@" 2> $o/bench_$name.err | grep '^{' | tail -1 > $o/bench_$name.json; }
b gat-lp_train --workload gat-lp --train --steps 32 --warmup 8
b emulated_world8 --workload mag240m-sharded --emulate-world 8 --no-cpu-baseline --steps 256
GIGL_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 20 --warmup 5 --shard-scale 0.01 2> $o/bench_n2_shared_gpu.err | grep '^{' | tail -1 > $o/bench_n2_shared_gpu.json
python - <<P
import json, glob
for f in sorted(glob.glob("$o/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["metric"][:40], "value", "%.4g" % d["value"], "ms/step", round(d["ms_per_step"], 5), d["config"].get("autograd_driven_ms_per_step"))
    except Exception as e:
        print(f.split("/")[-1], "no line", e)
P
