#!/bin/bash
# On the GPU box: the full bench line (parity gate, CPU baseline, roofline, determinism self-check) of every non-headline workload, then the
# minibatch-size sweep.   bash tools/bench_all.sh v4   -> gpurun_out/bench_line_<workload>_v4.json, gpurun_out/sizes_v4.txt
V=${1:-vX}
cd $GRAFT_REPO_ROOT
for wl in cloth_hepi rigid2_empn rope_hepi rope_hepi_var rope_hepi_bf16; do
  python bench.py --workload $wl > gpurun_out/bench_line_${wl}_$V.json 2> gpurun_out/bench_${wl}_$V.err
  python - "$wl" "gpurun_out/bench_line_${wl}_$V.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); r = d["roofline"]
print(sys.argv[1], round(d["value"], 2), "steps/s", round(d["ms_per_step"], 3), "ms", r["bound"], round(r["frac"], 3),
      "parity", d["parity_gate"]["passed"] if d.get("parity_gate") else None, "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
done
bash tools/run_sizes.sh > gpurun_out/sizes_$V.txt 2>&1
cat gpurun_out/sizes_$V.txt
