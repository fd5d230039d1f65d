#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace.  Everything lands in gpurun_out/.
# usage: tools/gpu_check.sh [tag]
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/gpu.txt
echo "== pytest gpu" 
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s 2>&1 | tail -60 > $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
echo "== smoke"
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cat $OUT/bench.json
timeout 600 python bench.py --steps 6 --warmup 2 --single-launch 1 --cpu-images 0 --kernel-sweep 0 > $OUT/bench_single.json 2>> $OUT/bench.err; cat $OUT/bench_single.json
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-images 0 --kernel-sweep 0 > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1 )
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep only the small summaries (the raw trace can be large)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
