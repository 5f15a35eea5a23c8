#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" timeout 300 python scripts/debug/lp_plan_debug.py 2>&1 | grep -v Warn | grep "layer\|step" | cut -c1-330; }
run X=1
