#!/bin/bash
# On the GPU box: does the step get faster under SUSTAINED load?  (the rope lines of the final tree: three repeats at 17.0 ms, then 14.85)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sustained.txt
: > $OUT
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],2), 'steps/s  median', round(d['ms_per_step'],4), 'ms  repeats', [round(x,3) for x in d['repeats_ms_per_step']], 'calib', round(d['box_calibration']['mfma_tflops']), '->', round((d.get('box_calibration_after') or {}).get('mfma_tflops',0)))" >> $OUT; }
python bench.py --steps 100 --warmup 8 --repeats 21 --no-cpu-baseline --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | show rigid_4096_21x100
python bench.py --minibatch 512 --steps 400 --warmup 8 --repeats 21 --no-cpu-baseline --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | show rigid_512_21x400
python bench.py --workload rope_hepi_bf16 --steps 40 --warmup 8 --repeats 15 --no-cpu-baseline --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | show rope_bf16_15x40
python bench.py --workload cloth_hepi --steps 60 --warmup 8 --repeats 15 --no-cpu-baseline --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | show cloth_15x60
python bench.py --steps 20 --warmup 1500 --no-cpu-baseline --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | show rigid_4096_warm1500
cat $OUT
