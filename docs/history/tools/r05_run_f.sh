#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16_rope.py tests/test_gpu_determinism.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3
for wl in rope_hepi_bf16; do
python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_line_${wl}_r05h.json 2> gpurun_out/bench_${wl}_r05h.err
python - $wl <<'PY'
import json,sys
wl=sys.argv[1]
d=json.loads([l for l in open(f'gpurun_out/bench_line_{wl}_r05h.json') if l.startswith('{')][-1])
print(wl, round(d['value'],2), round(d['ms_per_step'],4), 'calib', round(d['box_calibration']['mfma_tflops']), d['loss'])
for k,v in list(d['roofline']['per_kernel_ms_per_step'].items())[:6]: print('  ',k,round(v,3))
PY
done
