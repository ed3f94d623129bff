#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16_rope.py tests/test_gpu_determinism.py tests/test_gpu_ops.py tests/test_gpu_step.py -x -q 2>&1 | tail -3
export GRL_ALLOW_DIAG_LIB=1
GRL_LIB=$PWD/_variants/lib_b16phase.so GRL_WORKLOAD=rope_hepi_bf16 python tools/edge_bwd16_phase.py 2>&1 | tail -12 | tee gpurun_out/edge_bwd16_phases_rope_bf16_r05d.txt
GRL_LIB=$PWD/_variants/lib_b16phase.so GRL_WORKLOAD=rigid_hepi python tools/edge_bwd16_phase.py 2>&1 | tail -12 | tee gpurun_out/edge_bwd16_phases_rigid_r05d.txt
unset GRL_ALLOW_DIAG_LIB
for wl in rope_hepi_bf16 rigid_hepi; do
python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_line_${wl}_r05d.json 2> gpurun_out/bench_${wl}_r05d.err
python - $wl <<'PY'
import json,sys
wl=sys.argv[1]
d=json.loads([l for l in open(f'gpurun_out/bench_line_{wl}_r05d.json') if l.startswith('{')][-1])
print(wl, round(d['value'],2), round(d['ms_per_step'],4), 'calib', round(d['box_calibration']['mfma_tflops']), d['parity_gate']['passed'] if d.get('parity_gate') else None, d['loss'])
for k,v in list(d['roofline']['per_kernel_ms_per_step'].items())[:6]: print('  ',k,round(v,3))
PY
done
