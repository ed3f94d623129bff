#!/bin/bash
# Run HERE (hipcc cross-compiles): build one libgrl_hip variant per set of flags under _variants/ (git-ignored, shipped by gpurun);
# `gpurun -- 'bash tools/run_variants.sh'` then times them on ONE box.   usage: bash tools/build_variants.sh name1 "-DA=0 -DB=1" name2 "..." ...
# A flag written @file.hip:-flag applies to that source file only (e.g. "@edge_conv16.hip:-fno-slp-vectorize"); it reaches BOTH builds of
# that file (fp32 and the -DGRL_PREC=1 twin).  Timing knock-outs (-DGRL_E16_NOGELU, ...: wrong results) compile only with -DGRL_DIAG.
# The jobs are geometry_rl_amd/hip.py's own (SOURCES + VARIANTS with FILE_FLAGS); a variant's flags come later on the command line and win.
# Variant libraries are linked WITHOUT the export map (diagnostic entry points such as grl_edge_bwd16_phase_read stay visible).
set -e
cd "$(dirname "$0")/.."
mkdir -p _variants
jobs_of() { python - <<'PY'
import sys
sys.path.insert(0, ".")
from geometry_rl_amd import hip
for s in hip.SOURCES:
    print(s, s + ".o", " ".join(hip.FILE_FLAGS.get(s, [])))
for s, fl, sfx in hip.VARIANTS:
    print(s, s + sfx + ".o", " ".join(hip.FILE_FLAGS.get(s, []) + fl))
PY
}
JOBS=$(jobs_of)
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( mkdir -p _variants/obj_$name && echo "$JOBS" | while read b obj extra; do
      common=""
      for t in $flags; do
        case $t in
          @$b:*) extra="$extra ${t#@$b:}" ;;
          @*) ;;
          *) common="$common $t" ;;
        esac
      done
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $extra $common -c geometry_rl_amd/csrc/$b -o _variants/obj_$name/$obj || exit 1
    done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/lib_$name.so _variants/obj_$name/*.o && rm -rf _variants/obj_$name && echo "built $name [$flags]" ) &
  if (( $(jobs -r | wc -l) >= 3 )); then wait -n; fi
done
wait
ls -la _variants
