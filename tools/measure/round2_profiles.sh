# What profiles/r02/* are made with: GPU tests, default bench, bench under rocprofv3 --kernel-trace --stats, and the PMC passes of
# the seed kernel (separate runs, counters only).  Outputs under gpurun_out/r02/; copy what is to be judged into profiles/r02/.
OUT=gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { tail -20 $OUT/gpu_tests.log; exit 1; }
tail -1 $OUT/gpu_tests.log
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log || { tail -5 $OUT/bench_default.log; exit 1; }
grep "\[bench\]" $OUT/bench_default.log
rm -rf $OUT/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log || exit 1
cp $(ls $OUT/prof/*/*kernel_stats.csv | tail -1) $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
  rm -rf $OUT/pmc_$C
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$C.json 2> $OUT/pmc_$C.log || { echo "pmc pass $C failed"; tail -3 $OUT/pmc_$C.log; exit 1; }
done
T=$(python3 -c "import json;print(json.load(open('$OUT/bench_default.json'))['config']['index']['layout_threads'])")
python3 tools/pmc_seed_r02.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_EA0_RDREQ_sum $OUT --steps 3 --warmup 1 --layout-threads $T
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
  f=$(ls $OUT/pmc_$C/*/*counter_collection.csv | tail -1); head -1 $f > $OUT/pmc_$C.csv; grep "k_seed" $f >> $OUT/pmc_$C.csv; rm -rf $OUT/pmc_$C
done
