#!/bin/bash
# Run HERE (hipcc cross-compiles): build one libgrl_hip variant per set of -D flags under _variants/ (git-ignored, shipped by gpurun);
# `gpurun -- 'bash tools/run_variants.sh'` then times them on ONE box.   usage: bash tools/build_variants.sh name1 "-DA=0 -DB=1" name2 "..." ...
set -e
cd "$(dirname "$0")/.."
mkdir -p _variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $flags -shared -o _variants/lib_$name.so geometry_rl_amd/csrc/*.hip && echo "built $name [$flags]" ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la _variants
