# PMC passes of the seed-lookup kernel on the GRCh38 stand-in (seed stage only): FETCH_SIZE, WRITE_SIZE, L2 hit/miss and SQ wait
# counters, each in its own rocprofv3 run (no trace domains beside --pmc).  Output: gpurun_out/$1/pmc_*.csv (rows of lnr::k_seed*).
OUT=gpurun_out/${1:-seedpmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/raw_$tag
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/raw_$tag -- python3 tools/grch38_probe.py --seed-only --batches 1 ${PROBE_ARGS} > $OUT/run_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $OUT/run_$tag.log; continue; }
  f=$(ls $OUT/raw_$tag/*/*counter_collection.csv | tail -1)
  head -1 $f > $OUT/pmc_$tag.csv; grep "k_seed" $f >> $OUT/pmc_$tag.csv
  python3 - "$OUT/pmc_$tag.csv" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(k[0], k[1], "launches", len(v), "last", v[-1], "mean", sum(v) / len(v))
PY
  rm -rf $OUT/raw_$tag
done
