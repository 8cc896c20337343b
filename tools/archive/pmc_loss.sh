#!/bin/bash
# PMC counters of the loss kernels (separate passes, kernel trace only): instruction mix and LDS bank conflicts.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pl_$tag -o loss -- python $R/tools/loss_only.py > /tmp/pl_$tag.log 2>&1
  db=$(find /tmp/pl_$tag -name '*.db' | head -1)
  [ -n "$db" ] && python $R/profiles/summarize_rocprof.py pmc $db | grep -i "ssim\|counter" | cut -c1-140 || { echo "no db for $set"; tail -3 /tmp/pl_$tag.log; }
done
