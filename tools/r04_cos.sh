cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( echo "# tools/grad_cos_report.py cfg1 cfg2 cfg3 cfg4: cosine of every gradient slice held in tests/golden/full_*.npz (fp32 reference gradients of the"
  echo "# HF model on the same batch) with this framework's gradient, all-fp16 training mode (default) and the bf16-operand mode of rounds 1-3"
  for amp in fp16 bf16; do CLDRD_AMP=$amp python3 tools/grad_cos_report.py cfg1 cfg2 cfg3 cfg4 2>&1 | grep -v amdgpu; done ) > gpurun_out/r04/grad_cosines.txt
cat gpurun_out/r04/grad_cosines.txt
