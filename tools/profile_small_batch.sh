#!/bin/bash
# rocprofv3 passes over the small-batch tower (run on the GPU box, from the repo root):
#   tools/profile_small_batch.sh TAG N        (N boards per azr_nn_predict call, 55 calls)
# -> gpurun_out/prof_TAG/: kernel_stats.csv (kernel-trace + stats) and pmc_summary.csv (each --pmc set in a run of its own)
set -u
TAG=$1; N=$2
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="tools/small_batch_probe.py $N"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/probe_under_rocprof.txt 2> $OUT/stats.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.err
python3 tools/pmc_summary.py $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT
