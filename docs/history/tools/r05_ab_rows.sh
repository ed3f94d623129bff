#!/bin/bash
# A/B on ONE box: how the plain-bf16 fused edge backward fetches the source node's rows (GRL_B16_ROWS 0 / 1), rope workload, alternating
cd $GRAFT_REPO_ROOT
GRL_VARIANT_ARGS="--workload rope_hepi_bf16 --steps 20 --warmup 4 --pool 8 --repeats 3 --no-parity-gate" GRL_VARIANT_ROUNDS="1 2 3" bash tools/run_variants.sh 2>&1 | tee gpurun_out/r05_ab_rows.txt
