cd $GRAFT_REPO_ROOT
for ph in 0 100 200 300 400 500; do echo "== CLDRD_GEMM_PHASE=$ph"; CLDRD_GEMM_PHASE=$ph python tools/epi_ablate.py 0 2>&1 | grep -v amdgpu | cut -c1-80; done
