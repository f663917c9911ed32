# Final measurement set of a round (run on the GPU box from the repo root; gpurun_out/final is merged back)
set +e
export TMPDIR=/tmp
R=$PWD
F=$R/gpurun_out/final
mkdir -p $F
bash tools/collect_traffic.sh > $F/traffic.log 2>&1
cp gpurun_out/traffic/gemm_hbm_traffic.json $F/ 2>/dev/null
( time timeout 900 python bench.py > $F/bench_default.json 2> $F/bench_default.err ) 2> $F/bench_default.time
timeout 300 python bench.py --precision fast --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $F/bench_c3_fast.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $F/bench_c3_bf16.json 2>/dev/null
timeout 300 python bench.py --dtype fp8 --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $F/bench_c3_fp8.json 2>/dev/null
timeout 600 python tools/bench_decode.py --tiles 1 --batch 1 8 --tokens 32 > $F/decode_short.txt 2>/dev/null
cd /tmp
for mode in lo4 fast; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $F/prof_$mode -o $mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast-line --no-other-configs --precision $mode > $F/prof_${mode}_bench.json 2> $F/prof_$mode.err
  f=$(find $F/prof_$mode -name "*kernel_stats.csv" | head -1); cp "$f" $F/bench_c3_${mode}_kernel_stats.csv; rm -rf $F/prof_$mode
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $F/pmc_lo4 -- python $R/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-fast-line --no-other-configs > /dev/null 2> $F/pmc_lo4.err
cd $R
f=$(find $F/pmc_lo4 -name "*counter_collection.csv" | head -1); python tools/mfma_util.py "$f" > $F/mfma_util_lo4.txt 2>&1; rm -rf $F/pmc_lo4
timeout 900 python tools/stress_repro.py 40 lo4 > $F/race_screen.txt 2>&1
timeout 600 python tools/stress_repro.py 20 >> $F/race_screen.txt 2>&1
head -14 $F/mfma_util_lo4.txt | cut -c1-180
for f in $F/bench_*.json; do echo "$f: $(cut -c1-300 $f)"; done
cat $F/race_screen.txt | tail -8
