#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06m; mkdir -p $o
timeout 5000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $o/pytest_gpu_tail.txt
cat $o/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
