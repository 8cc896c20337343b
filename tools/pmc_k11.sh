# SQ counters of blend_backward_compact_kernel for two builds of the library (libfgs_hip_ref.so = the commit before, libfgs_hip.so = current) on
# S2 (shift 0.0) and the layered scene (shift -3.0): instruction mix, issue / wait cycles, LDS conflicts. usage: bash tools/pmc_k11.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; cd /tmp; export TMPDIR=/tmp
for shift in -3.0 0.0; do for lib in libfgs_hip_ref.so libfgs_hip.so; do
  echo "== $lib shift $shift"
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    rm -rf /tmp/pk; FGS_HIP_LIBRARY=$R/faster-gaussian-splatting_amd/$lib rocprofv3 --kernel-trace --pmc $pass -d /tmp/pk -o p -- python $R/tools/layered_step.py $shift > /tmp/pk.log 2>&1
    db=$(find /tmp/pk -name '*.db' | head -1)
    python $R/profiles/summarize_rocprof.py pmc $db | grep "blend_backward_compact" | awk '{printf "  %-24s %16.0f\n", $2, $4}'
    python $R/profiles/summarize_rocprof.py stats $db | grep "blend_backward_compact" | awk '{print "  avg_us", $4}'
  done
done; done
