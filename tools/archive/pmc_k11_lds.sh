cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; cd /tmp; export TMPDIR=/tmp
for lib in libfgs_hip_ref.so libfgs_hip.so; do
  FGS_HIP_LIBRARY=$R/faster-gaussian-splatting_amd/$lib rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d /tmp/pk_$lib -o p -- python $R/tools/layered_step.py 0.0 > /tmp/pk_$lib.log 2>&1
  db=$(find /tmp/pk_$lib -name '*.db' | head -1); echo "== $lib"; python $R/profiles/summarize_rocprof.py pmc $db | grep "blend_backward_compact" | awk '{printf "  %-24s %14.0f\n", $2, $4}'; python $R/profiles/summarize_rocprof.py stats $db | grep "blend_backward_compact" | awk '{print "  avg_us", $4}'
done
