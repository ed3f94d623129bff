#!/bin/bash
cd $GRAFT_REPO_ROOT
export GRL_ALLOW_DIAG_LIB=1
GRL_LIB=$PWD/_variants/lib_b16phase.so GRL_WORKLOAD=rope_hepi_bf16 python tools/edge_bwd16_phase.py 2>&1 | tail -11 | tee gpurun_out/edge_bwd16_phases_rope_bf16_r05c.txt
GRL_LIB=$PWD/_variants/lib_b16phase.so GRL_WORKLOAD=rigid_hepi python tools/edge_bwd16_phase.py 2>&1 | tail -11 | tee gpurun_out/edge_bwd16_phases_rigid_r05c.txt
for lib in product nogather nogelu; do
  if [ $lib = product ]; then L=$PWD/geometry_rl_amd/libgrl_hip.so; else L=$PWD/_variants/lib_$lib.so; fi
  GRL_BENCH_NO_SELFCHECK=1 GRL_LIB=$L python bench.py --workload rope_hepi_bf16 --steps 20 --warmup 4 --pool 8 --repeats 3 --no-parity-gate 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_ms_per_step']
print('$lib'.ljust(10), 'steps/s %7.2f' % d['value'], ' '.join('%s %.3f' % (n[:14], k[n]) for n in list(k)[:6]))"
done
