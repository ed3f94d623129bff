#!/bin/bash
# Build-variant check on the GPU box: run-to-run determinism of the MFMA ops for each set of -D flags.
cd $GRAFT_REPO_ROOT
for v in "-DGRL_FWD_MAX_BLOCKS=256"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $v -shared -o /tmp/libv.so geometry_rl_amd/csrc/*.hip 2>&1 | grep error
  echo "== variant: [$v]"
  GRL_REPS=10 GRL_LIB=/tmp/libv.so python tools/det_check_all.py 2>&1 | grep -v "dgamma\|dbeta\|dW3\|db3\|dW4\|db4" | tail -30
done
