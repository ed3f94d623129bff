#!/bin/bash
# Run HERE (needs git + hipcc, no GPU): snapshot of the committed tree under _ab/base (git-ignored, but shipped by gpurun) with its own
# library, so that `gpurun -- 'bash tools/ab_bench.sh [bench args]'` can compare it with the working tree on ONE box.
set -e
cd "$(dirname "$0")/.."
rm -rf _ab/base && mkdir -p _ab/base
git archive "${1:-HEAD}" | tar -x -C _ab/base
(cd _ab/base && python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1)
ls -la _ab/base/geometry_rl_amd/libgrl_hip.so
