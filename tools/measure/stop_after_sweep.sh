# job-kernel time on the GRCh38 stand-in with every read leaving after phase k (LNR_STOP_AFTER): throughput cost of the phases
# under full concurrency.  2 binning, 3 radix sort, 13 list filter, 4 x-sort, 5 DP, 6 traceback, 7 block gather+prefilter,
# 10 block scores+sort, 11 block DP, 8 block traceback+rewrite, 9 window filter, 0 everything.
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
for k in 2 3 13 4 5 6 7 10 11 8 9 0; do
  LNR_STOP_AFTER=$k timeout -k 10 200 python tools/grch38_probe.py --batches 1 ${PROBE_ARGS} > $OUT/stop_$k.log 2>&1 || { echo "k=$k failed"; tail -3 $OUT/stop_$k.log; exit 1; }
  echo "stop_after $k: $(grep 'rep 1 batch 0' $OUT/stop_$k.log | sed 's/.*| ms/ms/')"
done
