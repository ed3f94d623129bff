#!/bin/bash
# On the GPU box: (1) the standalone reproducer of the MFMA / LDS operand hazard (tools/ubench/mfma_lds_hazard.hip, built into _variants/ by
# the caller), (2) the in-situ reproducer: the library built with -DGRL_FENCED_2W=false (unfenced MFMA groups at two waves per SIMD) under
# tools/det_check_all.py.  Output -> gpurun_out/hazard_check.txt
cd $GRAFT_REPO_ROOT
{
  echo "== standalone chain (tools/ubench/mfma_lds_hazard.hip)"
  ./_variants/mfma_lds_hazard 30
  echo "== in situ: library with unfenced groups (-DGRL_FENCED_2W=false), tools/det_check_all.py, 12 repetitions"
  GRL_REPS=12 GRL_LIB=$PWD/_variants/lib_unfenced.so python tools/det_check_all.py 2>&1 | tail -12
  echo "== in situ: shipped library (fenced), same check"
  GRL_REPS=12 python tools/det_check_all.py 2>&1 | tail -4
} > gpurun_out/hazard_check.txt 2>&1
cat gpurun_out/hazard_check.txt
