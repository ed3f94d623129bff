#!/bin/bash
# Run HERE (hipcc cross-compiles): build one libgrl_hip variant per set of flags under _variants/ (git-ignored, shipped by gpurun);
# `gpurun -- 'bash tools/run_variants.sh'` then times them on ONE box.   usage: bash tools/build_variants.sh name1 "-DA=0 -DB=1" name2 "..." ...
# A flag written @file.hip:-flag applies to that source file only (e.g. "@edge_conv16.hip:-fno-slp-vectorize").
# Timing knock-outs (-DGRL_E16_NOGELU, -DGRL_KNOCK_STAGE, ...: wrong results) compile only together with -DGRL_DIAG (csrc/grl_common.h).
# The per-file flags of geometry_rl_amd/hip.py FILE_FLAGS are applied first (a variant's flags come later on the command line and win).
set -e
cd "$(dirname "$0")/.."
mkdir -p _variants
base_flags() { python - "$1" <<'PY'
import sys, re, ast
src = open("geometry_rl_amd/hip.py").read()
m = re.search(r"FILE_FLAGS = (\{.*?\})\n", src, re.S)
print(" ".join(ast.literal_eval(m.group(1)).get(sys.argv[1], [])))
PY
}
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( mkdir -p _variants/obj_$name && for f in geometry_rl_amd/csrc/*.hip; do
      b=$(basename $f); extra=$(base_flags $b); common=""
      for t in $flags; do
        case $t in
          @$b:*) extra="$extra ${t#@$b:}" ;;
          @*) ;;
          *) common="$common $t" ;;
        esac
      done
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $extra $common -c $f -o _variants/obj_$name/$b.o || exit 1
    done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/lib_$name.so _variants/obj_$name/*.o && rm -rf _variants/obj_$name && echo "built $name [$flags]" ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la _variants
