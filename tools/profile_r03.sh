#!/bin/bash
# Run on the GPU box (gpurun): kernel traces of the training step (eager, so every kernel shows by name) and of the decode; outputs
# under gpurun_out/prof_r03/.  Each rocprofv3 pass is its own short run (counters never combined with traces).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_train -o k --output-format csv -- python3 tools/train_only.py 128 10 0 0 > $OUT/kt_train.log 2>&1
python3 tools/rocpd_to_csv.py stats $(ls $OUT/kt_train/*/*.db $OUT/kt_train/*.db 2>/dev/null | head -1) > $OUT/train_step_b128_kernel_stats.csv 2>/dev/null || true
if [ "$1" = "all" ]; then
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_train_dropout -o k --output-format csv -- python3 tools/train_only.py 128 10 1 0 > $OUT/kt_train_dropout.log 2>&1
python3 tools/rocpd_to_csv.py stats $(ls $OUT/kt_train_dropout/*/*.db $OUT/kt_train_dropout/*.db 2>/dev/null | head -1) > $OUT/train_step_b128_dropout_kernel_stats.csv 2>/dev/null || true
timeout -s KILL 90 rocprofv3 --kernel-trace --stats -d $OUT/kt -o k --output-format csv -- python3 tools/decode_only.py 128 auto 100 > $OUT/kt.log 2>&1
python3 tools/rocpd_to_csv.py stats $(ls $OUT/kt/*/*.db $OUT/kt/*.db 2>/dev/null | head -1) > $OUT/decode_b128_basic_kernel_stats.csv 2>/dev/null || true
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_bench -o k --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-scalings > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python3 tools/rocpd_to_csv.py stats $(ls $OUT/kt_bench/*/*.db $OUT/kt_bench/*.db 2>/dev/null | head -1) > $OUT/bench_b128_basic_kernel_stats.csv 2>/dev/null || true
fi
ls $OUT
echo profiled
