cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in ${LIBS:-libfgs_hip.so libfgs_hip_ref.so}; do for r in 1 2 3 4 5 6; do
  FGS_HIP_LIBRARY=$PWD/faster-gaussian-splatting_amd/$lib timeout 120 python -m pytest tests/test_gpu_graph.py -m gpu -x -q 2>&1 | grep -E "passed|failed|assert \(" | tr '\n' ' ' | sed "s/^/$lib $r: /"; echo
done; done
