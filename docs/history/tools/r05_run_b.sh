#!/bin/bash
# round 5, second GPU call: bf16 edge kernels with resident / layer-ahead fragments -- parity + determinism tests, bench, phases, 16-bit VALU rates
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_rope.py tests/test_gpu_determinism.py tests/test_gpu_ops.py -x -q 2>&1 | tail -4
python bench.py --workload rope_hepi_bf16 --no-cpu-baseline > gpurun_out/bench_line_rope_hepi_bf16_r05b.json 2> gpurun_out/bench_rope_bf16_r05b.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_line_rope_hepi_bf16_r05b.json') if l.startswith('{')][-1])
print('rope bf16', d['value'], d['ms_per_step'], d['box_calibration']['mfma_tflops'], d['loss'])
for k,v in list(d['roofline']['per_kernel_ms_per_step'].items())[:8]: print('  ',k,round(v,3))
PY
export GRL_ALLOW_DIAG_LIB=1
GRL_LIB=$PWD/_variants/lib_b16phase.so GRL_WORKLOAD=rope_hepi_bf16 python tools/edge_bwd16_phase.py 2>&1 | tail -9 | tee gpurun_out/edge_bwd16_phases_rope_bf16_r05b.txt
GRL_LIB=$PWD/_variants/lib_b16phase.so GRL_WORKLOAD=rigid_hepi python tools/edge_bwd16_phase.py 2>&1 | tail -9 | tee gpurun_out/edge_bwd16_phases_rigid_r05b.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/ubench/valu16_rates.hip -o /tmp/valu16_rates && /tmp/valu16_rates | tee gpurun_out/valu16_rates.txt
