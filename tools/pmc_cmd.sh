#!/bin/bash
# PMC passes (separate passes; --pmc only with --kernel-trace) over an arbitrary python script.  On the GPU box, from the repo root:
#   bash tools/pmc_cmd.sh <tag> tools/mlp_bwd_bench.py [args]      -> gpurun_out/pmc_<tag>_table.txt, gpurun_out/pmc_<tag>_summary.json
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmcA -- python3 $R/"$@" > $O/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmcB -- python3 $R/"$@" > $O/pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcC -- python3 $R/"$@" > $O/pmcC.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcD -- python3 $R/"$@" > $O/pmcD.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $O/pmcE -- python3 $R/"$@" > $O/pmcE.log 2>&1
ls $O/pmcE/*/*counter_collection.csv > /dev/null 2>&1 || { rm -rf $O/pmcE; rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmcE -- python3 $R/"$@" > $O/pmcE.log 2>&1; }
cd $R
python tools/pmc_report.py $O $O/../pmc_${TAG}_summary.json > gpurun_out/pmc_${TAG}_table.txt
find $O -name "*.csv" -size +200k -delete   # keep the merged scratch under the 64 MiB limit
cat gpurun_out/pmc_${TAG}_table.txt
