#!/bin/bash
# On the GPU box: the final tree once more -- GPU suite, the data-parallel program beside the one-rank program, the driver's own calls.
cd $GRAFT_REPO_ROOT
V=${1:-v5}
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite_$V.txt 2>&1; echo "suite rc $?" >> gpurun_out/gpu_suite_$V.txt
tail -4 gpurun_out/gpu_suite_$V.txt
timeout 900 bash tools/dp_stats_ab.sh > gpurun_out/dp_plan_$V.txt 2>&1
head -14 gpurun_out/dp_plan_$V.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>/dev/null | grep "^{" | tail -1 > gpurun_out/bench_line_${V}_driver_path.json
python -c "
import json; d=json.load(open('gpurun_out/bench_line_${V}_driver_path.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d.get('box_calibration',{}).get('mfma_tflops'))"
