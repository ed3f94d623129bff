#!/bin/bash
# Build-variant check on the GPU box: run-to-run determinism of the MFMA ops + step throughput for each set of -D flags.
cd $GRAFT_REPO_ROOT
for v in "-DGRL_FWD_MAX_BLOCKS=256" "-DGRL_FWD_MAX_BLOCKS=512"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $v -shared -o /tmp/libv.so geometry_rl_amd/csrc/*.hip 2>&1 | grep error
  echo "== variant: [$v]"
  GRL_REPS=12 GRL_LIB=/tmp/libv.so python tools/det_check_all.py 2>&1 | grep -v "dgamma\|dbeta\|dW\|db[0-9]" | tail -6
  GRL_LIB=/tmp/libv.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps/s', round(d['value'],2), {k:round(v,3) for k,v in list(d['roofline']['per_kernel_ms_per_step'].items())[:7]})"
done
