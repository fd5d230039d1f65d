#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, probes, rocprof.  Everything lands in gpurun_out/<tag>/.
# usage: tools/gpu_check.sh <tag> [steps...]   steps: tests smoke bench probe rocprof pmc dimvariants widened ...
TAG=${1:-r1}; shift
STEPS=${@:-tests smoke bench rocprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $OUT/host.txt
for step in $STEPS; do
case $step in
tests)
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
  grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -30 ;;
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
bench)
  timeout 900 python bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
  timeout 600 python bench.py --steps 6 --warmup 2 --single-launch 1 --cpu-images 0 --kernel-sweep 0 > $OUT/bench_single.json 2>> $OUT/bench.err; cat $OUT/bench_single.json ;;
batches)
  for b in 64 125 250; do timeout 600 python bench.py --steps 3 --warmup 1 --batch $b --cpu-images 0 --kernel-sweep 0 2>> $OUT/bench.err | tee -a $OUT/bench_batches.json; done ;;
kernels)
  timeout 600 python tools/kernel_bench.py > $OUT/kernel_bench.json 2> $OUT/kernel_bench.err; cat $OUT/kernel_bench.json; tail -3 $OUT/kernel_bench.err ;;
timvariants)
  for v in 0 1 2 3; do TA_TIM_VARIANT=$v timeout 300 python tools/kernel_bench.py 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('variant=$v', {k:v for k,v in d.items() if 'tim' in k or 'dim' in k})"; done | tee $OUT/tim_variants.txt
  for v in 2 3; do TA_TIM_VARIANT=$v timeout 300 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "tim or dim" -p no:cacheprovider 2>&1 | tail -2; done ;;
dimvariants)
  # first thing to run next round: parity of the lane-per-column DIM kernels on the device, then their timing
  TA_DIM_FWD_VARIANT=2 TA_DIM_BWD_VARIANT=1 timeout 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_attacks.py -q -m gpu -k "dim or dts" -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/dim_variants_pytest.txt
  for v in "0 0" "2 0" "0 1" "2 1"; do set -- $v; TA_DIM_FWD_VARIANT=$1 TA_DIM_BWD_VARIANT=$2 timeout 120 python tools/dim_time.py; done 2>&1 | tee $OUT/dim_variants.txt
  TA_XCD_MAJOR_TILES=1 TA_DIM_FWD_VARIANT=2 TA_DIM_BWD_VARIANT=1 timeout 120 python tools/dim_time.py 2>&1 | sed "s/^/xcd-major /" | tee -a $OUT/dim_variants.txt ;;
widened)
  timeout 300 python tools/tim_microbench.py 2>&1 | tail -5 | tee $OUT/tim_separable.txt
  timeout 600 python -m pytest tests/test_zz_hip_widened.py -q -m gpu -s -p no:cacheprovider 2>&1 | tail -12 | tee $OUT/widened_pytest.txt ;;
freq)
  # SSM (20 spectrum views per iteration) with the rocFFT DCT pair and with the GEMM form
  for v in 0 1; do TA_DCT_GEMM=$v timeout 600 python bench.py --attack ssm --batch 16 --steps 1 --warmup 1 --cpu-images 0 --kernel-sweep 0 2>> $OUT/bench.err | tail -1 | sed "s/^/TA_DCT_GEMM=$v /" | tee -a $OUT/bench_ssm_b16.txt; done ;;
k2sweep)
  timeout 300 tools/bin/k2_sweep > $OUT/k2_sweep.txt 2>&1; cat $OUT/k2_sweep.txt ;;
fast)
  TA_FOLD_BN=1 TA_CHANNELS_LAST=1 timeout 600 python bench.py --steps 4 --warmup 2 --batch 125 --cpu-images 0 --kernel-sweep 0 2>> $OUT/bench.err | tee $OUT/bench_fast125.json
  TA_FOLD_BN=1 TA_CHANNELS_LAST=1 timeout 600 python bench.py --steps 6 --warmup 2 --batch 32 --cpu-images 0 --kernel-sweep 0 2>> $OUT/bench.err | tee $OUT/bench_fast32.json ;;
configs)
  timeout 600 python bench.py --attack sia --batch 16 --steps 2 --warmup 1 --cpu-images 0 --kernel-sweep 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_sia_b16.json
  timeout 600 python bench.py --attack dts --batch 32 --steps 3 --warmup 1 --cpu-images 0 --kernel-sweep 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_dts_b32.json
  timeout 900 python bench.py --attack vmifgsm --model vit_base_patch16_224 --batch 32 --steps 1 --warmup 1 --cpu-images 0 --kernel-sweep 0 --fold-bn 0 --channels-last 0 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_vmi_vit_b32.json
  timeout 600 python bench.py --attack ens --model resnet50,vgg16,inception_v3,vit_base_patch16_224 --batch 32 --steps 2 --warmup 1 --cpu-images 0 --kernel-sweep 0 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_ens4_b32.json ;;
probe)
  timeout 900 python tools/backbone_probe.py > $OUT/probe.jsonl 2> $OUT/probe.err; cat $OUT/probe.jsonl ;;
rocprof)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 8 --warmup 2 --cpu-images 0 --kernel-sweep 0 > $R/$OUT/rocprof.log 2>&1 )
  tail -1 $OUT/rocprof.log
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 "$f"; grep -E "ta::" "$f"; grep -v naive "$f" | head -12 | cut -c1-160)
  find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/tools/update_microbench.py > $R/$OUT/pmc_$c.log 2>&1 )
  done
  python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt ;;
esac
done
du -sh $OUT
