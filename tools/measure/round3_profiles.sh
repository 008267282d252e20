# What profiles/r03/* are made with.  bash tools/measure/round3_profiles.sh [tests|bench|stats|pmc|sq|phases ...] (default: all), outputs under
# gpurun_out/r03/; copy what is to be judged into profiles/r03/.  Every rocprofv3 run has the program directly behind `--`; counter passes are
# separate runs with --pmc only.
OUT=gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
WHAT="${@:-tests bench stats pmc sq phases gapprof}"
for w in $WHAT; do case $w in
tests)
  timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { tail -20 $OUT/gpu_tests.log; exit 1; }
  tail -1 $OUT/gpu_tests.log ;;
bench)
  timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log || { tail -5 $OUT/bench_default.log; exit 1; }
  grep "\[bench\]" $OUT/bench_default.log | tail -12
  timeout -k 10 500 python3 bench.py --gap 50 --no-cli > $OUT/bench_gap50.json 2> $OUT/bench_gap50.log || { tail -5 $OUT/bench_gap50.log; exit 1; }
  timeout -k 10 500 python3 bench.py --workload ccs_sv --no-cli > $OUT/bench_ccs_sv.json 2> $OUT/bench_ccs_sv.log || { tail -5 $OUT/bench_ccs_sv.log; exit 1; }
  grep "\[bench\]" $OUT/bench_ccs_sv.log | tail -4 ;;
stats)
  rm -rf $OUT/prof
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cli --no-gap50 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log || exit 1
  cp $(ls $OUT/prof/*/*kernel_stats.csv | tail -1) $OUT/kernel_stats.csv; rm -rf $OUT/prof
  head -8 $OUT/kernel_stats.csv | cut -c1-150
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --gap 50 --steps 3 --warmup 1 --no-cpu-baseline --no-cli > $OUT/bench_gap50_under_rocprof.json 2> $OUT/bench_gap50_under_rocprof.log || exit 1
  cp $(ls $OUT/prof/*/*kernel_stats.csv | tail -1) $OUT/kernel_stats_gap50.csv; rm -rf $OUT/prof
  head -6 $OUT/kernel_stats_gap50.csv | cut -c1-150 ;;
pmc)
  for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
    rm -rf $OUT/pmc_$C
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-gap50 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.log || { echo "pmc pass $C failed"; tail -3 $OUT/pmc_$C.log; exit 1; }
  done
  T=$(python3 -c "import json;print(json.loads(open('$OUT/pmc_FETCH_SIZE.json').read().strip().splitlines()[-1])['config']['index']['layout_threads'])")
  python3 tools/pmc_seed_r02.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_EA0_RDREQ_sum $OUT --steps 3 --warmup 1 --layout-threads $T
  for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
    f=$(ls $OUT/pmc_$C/*/*counter_collection.csv | tail -1); head -1 $f > $OUT/pmc_$C.csv; grep "k_seed" $f >> $OUT/pmc_$C.csv; rm -rf $OUT/pmc_$C
  done ;;
sq)
  bash tools/measure/job_icache.sh > $OUT/job_sq_counters.txt 2>&1; tail -48 $OUT/job_sq_counters.txt; rm -rf gpurun_out/ic ;;
gapprof)
  # per-phase wave time of the gap stage and its slowest reads (-DLNR_GAP_DEVPROF build: python -m linear_amd.build gapprof -DLNR_GAP_DEVPROF)
  LNR_LIB=tools/_variants/gapprof.so timeout -k 10 400 python3 bench.py --gap 50 --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $OUT/bench_gap50_devprof.json 2> $OUT/gap_devprof_fused.txt || { tail -5 $OUT/gap_devprof_fused.txt; exit 1; }
  grep "gap prof" $OUT/gap_devprof_fused.txt | tail -45 | cut -c1-230 ;;
phases)
  timeout -k 10 300 python3 tools/prof_job_phases.py 100000 grch38 > $OUT/job_phases_grch38.log 2>&1; sed -n 2,16p $OUT/job_phases_grch38.log | cut -c1-220; rm -f gpurun_out/timeline_r*.npy ;;
esac; done
