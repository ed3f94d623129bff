#!/bin/bash
# Run HERE (hipcc cross-compiles): build one libgrl_hip variant per set of -D flags under _variants/ (git-ignored, shipped by gpurun);
# `gpurun -- 'bash tools/run_variants.sh'` then times them on ONE box.   usage: bash tools/build_variants.sh name1 "-DA=0 -DB=1" name2 "..." ...
set -e
cd "$(dirname "$0")/.."
mkdir -p _variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( mkdir -p _variants/obj_$name && for f in geometry_rl_amd/csrc/*.hip; do
      extra=""; { [ "$(basename $f)" = edge_conv16.hip ] || [ "$(basename $f)" = node_mlp16.hip ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form"   # as geometry_rl_amd/hip.py FILE_FLAGS
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $extra $flags -c $f -o _variants/obj_$name/$(basename $f).o || exit 1
    done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/lib_$name.so _variants/obj_$name/*.o && rm -rf _variants/obj_$name && echo "built $name [$flags]" ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la _variants
