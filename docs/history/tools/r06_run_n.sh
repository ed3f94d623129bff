#!/bin/bash
# On the GPU box: what the driver runs at round end -- build() (must say prebuilt by hash), smoke(), the default bench line (traffic filled from the committed PMC summary).
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
cat BUILD_INFO.json | head -4
python bench.py > gpurun_out/bench_line_r06n.json 2> gpurun_out/bench_r06n.err; tail -c 600 gpurun_out/bench_line_r06n.json; echo
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_line_r06n.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['traffic'], r['traffic_source'], r['frac'], r.get('frac_alg_3xfwd'))
PY
