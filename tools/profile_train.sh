#!/bin/bash
# rocprofv3 kernel table of the optimiser step (run on the GPU box, from the repo root):  tools/profile_train.sh TAG [train_bench args]
# -> gpurun_out/prof_train_TAG/kernel_stats.csv (+ the bench line)
set -u
TAG=$1; shift
OUT=gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/train_bench.py "$@" > $OUT/bench.txt 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
grep blocks= $OUT/bench.txt
