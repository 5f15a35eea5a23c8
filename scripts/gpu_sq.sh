#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/gpu_sq.sh <tag> [kernel regex] [bench args...]
# SQ instruction-mix counters of one kernel (default: the sampler's expand_kernel) in a short single-stream,
# eager-launch bench run; passes of <= 4 counters each, --kernel-trace only.  scripts/sq_summary.py folds them into
# gpurun_out/sq_<tag>.json: per dispatch shape (grid), mean counter per dispatch.
set -u
tag=$1; shift
re=${1:-expand_kernel}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
            "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" \
            "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
  out=gpurun_out/sq_${tag}/p$i
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "$re" -f csv -d "$out" -o sq -- \
      python bench.py --no-live-pmc --no-emulated-sub --streams 1 --no-graph --steps 64 --min-rounds 2 --warmup 32 --timed-only "$@" \
      > gpurun_out/sq_${tag}_p$i.log 2>&1
  i=$((i+1))
done
python scripts/sq_summary.py "$tag"
find gpurun_out/sq_${tag} -name '*.csv' -size +4M -delete
