# What profiles/r01/v4_* were made with: GPU tests, default bench, bench under rocprofv3 --kernel-trace --stats, and the
# -DLNR_PROF phase profile.  Outputs under gpurun_out/v4/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/v4
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/v4/gpu_tests.log 2>&1 || { tail -20 gpurun_out/v4/gpu_tests.log; exit 1; }
tail -1 gpurun_out/v4/gpu_tests.log
timeout -k 10 400 python3 bench.py > gpurun_out/v4/bench_default.json 2> gpurun_out/v4/bench_default.log || { tail -5 gpurun_out/v4/bench_default.log; exit 1; }
cat gpurun_out/v4/bench_default.json
rm -rf gpurun_out/v4/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v4/prof -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/v4/bench_under_rocprof.json 2> gpurun_out/v4/bench_under_rocprof.log || exit 1
cat gpurun_out/v4/bench_under_rocprof.json
cp $(ls gpurun_out/v4/prof/*/*kernel_stats.csv | tail -1) gpurun_out/v4/kernel_stats.csv
head -12 gpurun_out/v4/kernel_stats.csv | cut -c1-160
timeout -k 10 300 python3 tools/prof_job_phases.py 100000 > gpurun_out/v4/phases.log 2>&1; grep -E "^==|biggest" gpurun_out/v4/phases.log | cut -c1-300
