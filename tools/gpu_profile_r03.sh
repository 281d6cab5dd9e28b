#!/bin/bash
# One gpurun call: bench line, kernel-trace stats, three PMC passes (SQ MFMA counters, FETCH_SIZE, WRITE_SIZE).
# usage (from the repo root on the GPU box): bash tools/gpu_profile_r03.sh <tag>
# (no `set -e`: rocprofv3 has been seen to crash in a static destructor AFTER it has written its output -- the files are
# complete, the exit code is not 0)
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --family-steps 0"
rocprofv3 --kernel-trace --stats --truncate-kernels --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
echo trace done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --truncate-kernels --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
echo pmc_sq done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --truncate-kernels --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
echo pmc_fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --truncate-kernels --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
echo pmc_write done
# keep what travels back small: counter CSVs can be large -> summarise on the box, keep the stats CSVs
python tools/pmc_report.py $OUT > $OUT/pmc_report.txt 2>&1 || true
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python profiles/summarize.py $T 60 > $OUT/last_step_summary.txt 2>&1 || true
python tools/trace_gaps.py $T >> $OUT/last_step_summary.txt 2>&1 || true
python tools/stream_timeline.py $T > $OUT/timeline_main.txt 2>&1 || true
python lab/steady_segments.py > $OUT/steady_segments.txt 2>&1 || true
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv || true
find $OUT -name '*counter_collection.csv' -size +20M -delete
find $OUT -name '*kernel_trace.csv' -size +20M -delete
du -sh $OUT
