#!/bin/bash
# SQ counters of the two blend kernels on the layered scene and on S2 (two passes of 8 SQ counters each)
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=gpurun_out
for shift in -3.0 0.0; do
  tag=$(echo $shift | tr -d '.-')
  ( cd /tmp; export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/$O/pmcb_a$tag -o p -- python $R/tools/layered_step.py $shift > $R/$O/pmcb_a$tag.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d $R/$O/pmcb_b$tag -o p -- python $R/tools/layered_step.py $shift > $R/$O/pmcb_b$tag.log 2>&1 )
  for k in a b; do db=$(find $O/pmcb_$k$tag -name '*.db' | head -1); python profiles/summarize_rocprof.py pmc $db | grep -E "blend_kernel<true>|blend_backward_compact|kernel  " > $O/pmc_blend_${k}_shift$tag.txt; python profiles/summarize_rocprof.py stats $db | grep -E "blend_kernel<true>|blend_backward_compact|kernel  " >> $O/pmc_blend_${k}_shift$tag.txt; done
  find $O/pmcb_a$tag $O/pmcb_b$tag -name '*.db' -delete
done
cat $O/pmc_blend_*_shift*.txt
