#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "-DGRL_DBG_HALF_IDLE" "-DGRL_FWD_WAVES=4"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $v -shared -o /tmp/libv.so geometry_rl_amd/csrc/*.hip 2>&1 | grep error
  echo "== variant: [$v]"
  GRL_LIB=/tmp/libv.so python tools/det_check2.py 2>&1 | grep "n bad"
done
