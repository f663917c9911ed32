set +e
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/final
bash tools/collect_traffic.sh > gpurun_out/final/traffic.log 2>&1
cp gpurun_out/traffic/gemm_hbm_traffic.json gpurun_out/final/ 2>/dev/null
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/final/bench_c3_f16.json 2> gpurun_out/final/bench_c3_f16.err
timeout 200 python bench.py --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/final/bench_c3_bf16.json 2>/dev/null
timeout 200 python bench.py --dtype fp8 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/final/bench_c3_fp8.json 2>/dev/null
timeout 200 python bench.py --dtype fp8 --fp8-attention 0 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/final/bench_c3_fp8_f16_attention.json 2>/dev/null
timeout 300 python bench.py --workload llava-c5 --graph-encode --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_c5_f16_graph.json 2>/dev/null
timeout 300 python bench.py --workload llava-c5 --dtype fp8 --graph-encode --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_c5_fp8_graph.json 2>/dev/null
timeout 300 python bench.py --workload idefics2-c4 --steps 10 --warmup 3 > gpurun_out/final/bench_idefics2_c4.json 2>/dev/null
timeout 200 python bench.py --images 1 --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/final/bench_c2_bf16.json 2>/dev/null
timeout 200 python bench.py --images 1 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/final/bench_c2_f16.json 2>/dev/null
timeout 200 python bench.py --images 1 --width 336 --height 336 --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/final/bench_c1_f16.json 2>/dev/null
timeout 600 python tools/bench_decode.py --tiles 1 --batch 1 2 4 8 16 --tokens 32 > gpurun_out/final/decode_short.txt 2>/dev/null
timeout 600 python tools/bench_decode.py --skip-single --tiles 42 --batch 1 8 --tokens 24 > gpurun_out/final/decode_long.txt 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_f16 -o f16 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/prof_f16_bench.json 2> $R/gpurun_out/final/prof_f16.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_fp8 -o fp8 -- python $R/bench.py --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/prof_fp8_bench.json 2> $R/gpurun_out/final/prof_fp8.err
cd $R
# decode, batch 1 (short context) and batch 8: per-kernel durations
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_dec1 -o dec1 -- python $R/tools/bench_decode.py --tiles 1 --tokens 64 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_dec8 -o dec8 -- python $R/tools/bench_decode.py --skip-single --tiles 1 --batch 8 --tokens 64 > /dev/null 2>&1
cd $R
for d in prof_dec1 prof_dec8; do f=$(find gpurun_out/final/$d -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/final/${d}_kernel_stats.csv; rm -rf gpurun_out/final/$d; done
for d in prof_f16 prof_fp8; do f=$(find gpurun_out/final/$d -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/final/${d}_kernel_stats.csv; rm -rf gpurun_out/final/$d; done
for f in gpurun_out/final/bench_*.json; do echo "$f: $(cut -c1-260 $f)"; done
head -12 gpurun_out/final/prof_fp8_kernel_stats.csv | cut -c1-200
# matrix-pipe utilisation per kernel (counters only: no tracing options beside --pmc)
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/final/pmc_f16 -- python $R/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > /dev/null 2> $R/gpurun_out/final/pmc_f16.err
cd $R
f=$(find gpurun_out/final/pmc_f16 -name "*counter_collection.csv" | head -1); python tools/mfma_util.py "$f" > gpurun_out/final/mfma_util_f16.txt 2>&1; rm -rf gpurun_out/final/pmc_f16
head -14 gpurun_out/final/mfma_util_f16.txt | cut -c1-180
