#!/bin/bash
# kernel trace of a few bench steps -> per-stream timeline of the last step (tools/stream_timeline.py)
# usage (repo root on the GPU box): bash tools/gpu_timeline.sh <tag> [ENV=VAL ...]
TAG=${1:-tl}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --truncate-kernels --output-format csv -d $OUT/trace -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-variants --family-steps 0 > $OUT/trace.log 2>&1
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python tools/stream_timeline.py $T > $OUT/timeline_main.txt 2>&1
python profiles/summarize.py $T 60 > $OUT/last_step_summary.txt 2>&1
python tools/trace_gaps.py $T 12 >> $OUT/last_step_summary.txt 2>&1
find $OUT -name '*kernel_trace.csv' -size +20M -delete
tail -3 $OUT/timeline_main.txt
