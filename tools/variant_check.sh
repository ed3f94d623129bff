#!/bin/bash
# Build-variant timing on the GPU box.
cd $GRAFT_REPO_ROOT
for v in "-DGRL_NONE" "-DGRL_DBG_FAKE_META"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $v -shared -o /tmp/libv.so geometry_rl_amd/csrc/*.hip 2>&1 | grep error
  echo "== variant: [$v]"
  GRL_LIB=/tmp/libv.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps/s', round(d['value'],2), {k:round(v,3) for k,v in list(d['roofline']['per_kernel_ms_per_step'].items())[:7]})"
done
