#!/bin/bash
# On the GPU box: the bench lines of the final tree on one box -- the headline line (what the driver runs), every other workload, the size sweep.
cd $GRAFT_REPO_ROOT
V=${1:-v6}
python bench.py > gpurun_out/bench_line_$V.json 2> gpurun_out/bench_$V.err
python - gpurun_out/bench_line_$V.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("rigid_hepi", round(d["value"], 2), "steps/s", round(d["ms_per_step"], 4), "ms  repeats", [round(x, 3) for x in d["repeats_ms_per_step"]], "frac", round(r["frac"], 3),
      round(r.get("frac_alg_3xfwd", 0), 3), "traffic", r["traffic"], "box", round(d["box_calibration"]["mfma_tflops"]), "lane_form", d.get("lane_form"))
PY
timeout 2400 bash tools/bench_all.sh $V
