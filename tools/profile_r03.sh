#!/bin/bash
# Run on the GPU box (gpurun): kernel traces of the training step (eager, so every kernel shows by name) and of the decode; outputs
# under gpurun_out/prof_r03/.  Each rocprofv3 pass is its own short run (counters never combined with traces).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
if [ "$1" != "pmc" ]; then
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_train -o k --output-format csv -- python3 tools/train_only.py 128 10 0 0 > $OUT/kt_train.log 2>&1
cp $OUT/kt_train/k_kernel_stats.csv $OUT/train_step_b128_kernel_stats.csv
fi
if [ "$1" = "all" ]; then
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_train_dropout -o k --output-format csv -- python3 tools/train_only.py 128 10 1 0 > $OUT/kt_train_dropout.log 2>&1
cp $OUT/kt_train_dropout/k_kernel_stats.csv $OUT/train_step_b128_dropout_kernel_stats.csv
timeout -s KILL 90 rocprofv3 --kernel-trace --stats -d $OUT/kt -o k --output-format csv -- python3 tools/decode_only.py 128 auto 100 > $OUT/kt.log 2>&1
cp $OUT/kt/k_kernel_stats.csv $OUT/decode_b128_basic_kernel_stats.csv
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_bench -o k --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-scalings > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $OUT/kt_bench/k_kernel_stats.csv $OUT/bench_b128_basic_kernel_stats.csv
fi
if [ "$1" = "pmc" ] || [ "$1" = "all" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 90 rocprofv3 --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python3 tools/decode_only.py 128 auto 25 > $OUT/pmc_$c.log 2>&1
done
python3 tools/traffic_json.py $OUT > $OUT/traffic_b128_basic.json
fi
ls $OUT
echo profiled
