# Instruction mix and wait cycles of EVERY kernel of the training iteration (tools/layered_step.py <shift>: four iterations; 0.0 = S2), per launch:
# which kernels sit on the scalar unit (SALU ~ VALU), on LDS, on waits. usage: bash tools/pmc_all_kernels.sh [shift]
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; cd /tmp; export TMPDIR=/tmp
shift_=${1:-0.0}
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pa; rocprofv3 --kernel-trace --pmc $pass -d /tmp/pa -o p -- python $R/tools/layered_step.py $shift_ > /tmp/pa.log 2>&1
  db=$(find /tmp/pa -name '*.db' | head -1)
  python $R/tools/pmc_all_kernels.py $db
done
