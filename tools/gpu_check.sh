#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, probes, rocprof.  Everything lands in gpurun_out/<tag>/.
# usage: tools/gpu_check.sh <tag> [steps...]   steps: tests smoke bench kernels configs pmc timpmc k2sweep rocprof ...
TAG=${1:-r2}; shift
STEPS=${@:-tests smoke bench rocprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $OUT/host.txt
for step in $STEPS; do
case $step in
tests)
  # the driver's tier (-m gpu, <= 900 s wanted against its 1200 s step limit), timed as the driver times it; then the 1000-image
  # forms of the two slowest ASR jobs (-m gpu_long)
  t0=$(date +%s)
  timeout ${TA_TESTS_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q --timeout 1200 -p no:cacheprovider -s --durations=25 > $OUT/pytest_gpu.log 2>&1
  echo "pytest -m gpu -x -q wall: $(( $(date +%s) - t0 )) s" | tee -a $OUT/pytest_gpu.log
  grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -30 ;;
newtests6)
  # round 6: the default (folded) loop form under replay, the deterministic hook-vs-folded comparison, the 256-image ASR jobs
  timeout 1200 python -m pytest tests/test_hip_configs.py tests/test_hip_attacks.py tests/test_hip_asr1000.py -q -m gpu -s -p no:cacheprovider --durations=12 \
      -k "config2 or config1 or fused_resnet or folded_loop or four_members or vmifgsm_vit or deterministic" 2>&1 | grep -v Warning | tee $OUT/newtests6_pytest.txt | tail -40 ;;
coldranks)
  # bench.py --gpus 2 on ONE device with fresh MIOpen databases: rank 0 warms up first vs both together (tools/cold_start_ranks.py)
  timeout 2400 python tools/cold_start_ranks.py --runs ${TA_COLD_RUNS:-one,one:warm,staged,together} 2> $OUT/cold_start_ranks.err | tee $OUT/cold_start_ranks.jsonl ;;
pool)
  # the stem pool pair (ta_maxpool3s2_fwd / _bwd_relu) against ATen on the device, and the surrogate around it
  timeout 900 python -m pytest tests/test_hip_configs.py -q -m gpu -s -p no:cacheprovider -k "stem_pool or maxpool or fused_glue" 2>&1 | grep -v Warning | tee $OUT/pool_pytest.txt | tail -15 ;;
stem)
  timeout 300 python tools/stem_microbench.py 2>&1 | tee $OUT/stem_microbench.txt ;;
asrlong)
  timeout 1200 python -m pytest tests -m gpu_long -q -p no:cacheprovider -s > $OUT/pytest_gpu_long.log 2>&1
  grep -E "passed|failed|images in" $OUT/pytest_gpu_long.log | tail -6 ;;
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
bench)
  timeout 900 python bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json ;;
batches)
  for b in 32 64 250; do timeout 600 python bench.py --steps 3 --warmup 1 --batch $b --cpu-images 0 --kernel-sweep 0 --literal-steps 0 2>> $OUT/bench.err | tee -a $OUT/bench_batches.json; done ;;
literal)
  # the reference's literal arrangement: batches of 32, NCHW, separate BatchNorm
  timeout 600 python bench.py --steps 6 --warmup 2 --batch 32 --fold-bn 0 --channels-last 0 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 2>> $OUT/bench.err | tee $OUT/bench_literal_b32.json ;;
kernels)
  timeout 600 python tools/kernel_bench.py > $OUT/kernel_bench.json 2> $OUT/kernel_bench.err; cat $OUT/kernel_bench.json; tail -3 $OUT/kernel_bench.err ;;
freq)
  # SSM (20 spectrum views per iteration): the transform is two launches of the fp32-MFMA kernel ta_dct_pair
  timeout 600 python bench.py --attack ssm --batch 16 --steps 1 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_ssm_b16.json
  timeout 300 python tools/spectrum_microbench.py 2>&1 | tail -8 | tee $OUT/spectrum_microbench.txt ;;
k2sweep)
  mkdir -p tools/bin; [ -x tools/bin/k2_sweep ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/k2_sweep.hip -o tools/bin/k2_sweep
  timeout 300 tools/bin/k2_sweep > $OUT/k2_sweep.txt 2>&1; cat $OUT/k2_sweep.txt ;;
configs)
  timeout 600 python bench.py --attack dts --batch 32 --steps 3 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_dts_b32.json
  timeout 600 python bench.py --attack ens --model resnet50,vgg16,inception_v3,vit_base_patch16_224 --batch 32 --steps 2 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_ens4_b32.json
  timeout 600 python bench.py --attack sia --batch 16 --steps 2 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_sia_b16.json ;;
vmi)
  timeout 900 python bench.py --attack vmifgsm --model vit_base_patch16_224 --batch 32 --steps 1 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 --fold-bn 0 --channels-last 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_vmi_vit_b32.json ;;
gpus1)
  # the self-launch path of bench.py --gpus N on a one-GPU box: N = 1 under torch.distributed.run (RCCL world of 1)
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_torchrun_n1.json
  timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 2>&1 | tail -1 | tee $OUT/bench_gpus2_refused.txt ;;
dispatch)
  timeout 120 python tools/update_dispatch_clock.py 2> $OUT/update_dispatch_clock.err | tee $OUT/update_dispatch_clock.json ;;
probe)
  timeout 900 python tools/backbone_probe.py > $OUT/probe.jsonl 2> $OUT/probe.err; cat $OUT/probe.jsonl ;;
ktrace)
  # rocprofv3's own durations of every kernel of the stand-alone bench (event timing adds the marker overhead)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace -o trace -- python $R/tools/kernel_bench.py > $R/$OUT/ktrace.log 2>&1 )
  f=$(find $OUT/ktrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_bench_kernel_stats.csv && head -40 "$f" | cut -c1-150
  find $OUT/ktrace -name "*kernel_trace.csv" -delete; find $OUT/ktrace -name "*.db" -delete ;;
e2e)
  # main.py end to end on 1000 synthetic PNGs: decode -> upload -> attack -> quantise -> download -> encode
  timeout 900 python tools/e2e_main.py 2> $OUT/e2e_main.err | tee $OUT/e2e_main.jsonl ;;
philox)
  mkdir -p tools/bin; [ -x tools/bin/philox_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/philox_rate.hip -o tools/bin/philox_rate
  timeout 120 tools/bin/philox_rate 2>&1 | tee $OUT/philox_rate.txt ;;
steady)
  # what the metric is made of: steady-state kernels of the default bench line.  MIOpen's find-db is warmed by a prior
  # process (its trial kernels stay out of the trace), then the trace is reduced ON THE BOX (tools/steady_trace.py)
  B=${TA_STEADY_BATCH:-125}; X=${TA_STEADY_EXTRA:-}
  timeout 600 python bench.py --steps 1 --warmup 1 --batch $B $X --cpu-images 0 --kernel-sweep 0 --literal-steps 0 > /dev/null 2>> $OUT/bench.err
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/steady_b$B -o trace -- python $R/bench.py --steps 4 --warmup 1 --batch $B $X --cpu-images 0 --kernel-sweep 0 --literal-steps 0 > $R/$OUT/steady_b$B.log 2>&1 )
  tail -1 $OUT/steady_b$B.log
  python tools/steady_trace.py $OUT/steady_b$B $OUT/steady_state_b$B.json 30 | tee $OUT/steady_state_b$B.txt
  f=$(find $OUT/steady_b$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/steady_state_b${B}_kernel_stats.csv
  find $OUT/steady_b$B -name "*kernel_trace.csv" -delete; find $OUT/steady_b$B -name "*.db" -delete ;;
rocprof)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 8 --warmup 2 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 > $R/$OUT/rocprof.log 2>&1 )
  tail -1 $OUT/rocprof.log
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 "$f"; grep -E "ta::" "$f"; grep -v naive "$f" | head -12 | cut -c1-160)
  find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete ;;
ens4)
  # configs[4] on one GPU at the reference's batch and at the per-GPU shard (ensemble MI-FGSM treats images independently)
  for b in 32 125; do
  timeout 900 python bench.py --attack ens --model resnet50,vgg16,inception_v3,vit_base_patch16_224 --batch $b --steps 2 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_ens4_b$b.json
  done ;;
dimclock)
  # where a DIM tile's cycles go: per-phase shader-clock cycles of the two lane-per-column kernels (a tuning build of dim.hip)
  timeout 300 python tools/dim_phase_clock.py 2>&1 | tee $OUT/dim_phase_clock.txt ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/tools/update_microbench.py > $R/$OUT/pmc_$c.log 2>&1 )
  done
  python tools/pmc_summary.py $OUT 125 --json $OUT/pmc_update_kernel.json > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
  find $OUT -name "*.db" -delete ;;
timpmc)
  ( cd /tmp && TA_N=160 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$OUT/timpmc -o pmc -- python $R/tools/tim_microbench.py > $R/$OUT/timpmc.log 2>&1 )
  python tools/pmc_kernels.py $OUT/timpmc | tee $OUT/timpmc_summary.txt
  find $OUT -name "*.db" -delete ;;
det)
  # TA_DETERMINISTIC=1: two processes write identical PNGs; what the switch costs on the bench line
  timeout 1500 python -m pytest tests/test_hip_attacks.py -q -m gpu -s -p no:cacheprovider -k "deterministic_mode" 2>&1 | grep -v Warning | tee $OUT/det_pytest.txt | tail -12
  TA_DETERMINISTIC=1 timeout 900 python bench.py --steps 4 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 2>> $OUT/bench.err | tee $OUT/bench_deterministic.json ;;
coldstart)
  timeout 1500 python tools/cold_start.py --modes ${TA_COLD_MODES:-immediate,immediate:warm,fast,fast:warm} --keep $OUT/miopen 2> $OUT/cold_start.err | tee $OUT/cold_start.jsonl
  du -sh $OUT/miopen/* 2>/dev/null ;;
dimab)
  # DIM pair + TIM: event timing, rocprofv3 durations and HBM counters at N = 160
  timeout 300 python tools/tim_microbench.py 2>&1 | tee $OUT/dim_tim_microbench.txt
  ( cd /tmp && TA_N=160 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/dimtrace -o trace -- python $R/tools/tim_microbench.py > $R/$OUT/dimtrace.log 2>&1 )
  f=$(find $OUT/dimtrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/dim_kernel_stats_n160.csv && grep -E "Name|dim_|dwconv" "$f" | cut -c1-200
  find $OUT/dimtrace -name "*kernel_trace.csv" -delete; find $OUT/dimtrace -name "*.db" -delete
  for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && TA_N=160 timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/dimpmc_$c -o pmc -- python $R/tools/tim_microbench.py > $R/$OUT/dimpmc_$c.log 2>&1 )
  python tools/pmc_kernels.py $OUT/dimpmc_$c | tee -a $OUT/dimpmc_summary.txt
  done
  find $OUT -name "*.db" -delete ;;
vmistack)
  # configs[3]: VMI-FGSM / ViT-B/16, k neighbour samples per surrogate evaluation
  for k in ${TA_VMI_KS:-1 5 10}; do
  TA_VMI_STACK=$k timeout 900 python bench.py --attack vmifgsm --model vit_base_patch16_224 --batch 32 --steps 1 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 --fold-bn 0 --channels-last 0 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_vmi_vit_b32_stack$k.json | cut -c1-400
  done ;;
trained)
  timeout 900 python -m pytest tests/test_hip_asr_trained.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep -v Warning | tee $OUT/asr_trained_pytest.txt | tail -30 ;;
dts)
  timeout 600 python bench.py --attack dts --batch 32 --steps 3 --warmup 1 --cpu-images 0 --kernel-sweep 0 --literal-steps 0 --kernel-times 1 2>> $OUT/bench.err | tail -1 | tee $OUT/bench_dts_b32.json | cut -c1-1500 ;;
esac
done
du -sh $OUT
