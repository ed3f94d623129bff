#!/bin/bash
# PMC passes over a few policy-update steps (separate passes; --pmc only with --kernel-trace). Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GRL_STEPS=2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/tools/profile_step.py > $R/gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/tools/profile_step.py > $R/gpurun_out/pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcC -- python3 $R/tools/profile_step.py > $R/gpurun_out/pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcD -- python3 $R/tools/profile_step.py > $R/gpurun_out/pmcD.log 2>&1
# pass E (round 5: where the wait + stall cycles go): MFMA / VALU co-execution, LDS / VMEM / scalar / misc issue activity.  If a counter name is
# unknown to this rocprofv3 the pass fails as a whole: pass F is the short list.
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $R/gpurun_out/pmcE -- python3 $R/tools/profile_step.py > $R/gpurun_out/pmcE.log 2>&1
ls $R/gpurun_out/pmcE/*/*counter_collection.csv > /dev/null 2>&1 || { rm -rf $R/gpurun_out/pmcE; rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcE -- python3 $R/tools/profile_step.py > $R/gpurun_out/pmcE.log 2>&1; }
echo done
