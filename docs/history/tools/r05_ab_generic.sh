#!/bin/bash
# A/B of every _variants/lib_*.so on ONE box, alternating rounds.  GRL_AB_WL: bench workload (default rope_hepi_bf16), GRL_AB_TAG: output tag
cd $GRAFT_REPO_ROOT
WL=${GRL_AB_WL:-rope_hepi_bf16}
GRL_VARIANT_ARGS="--workload $WL --steps 20 --warmup 4 --pool 8 --repeats 3 --no-parity-gate" GRL_VARIANT_ROUNDS="1 2 3" bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_ab_${GRL_AB_TAG:-x}.txt
