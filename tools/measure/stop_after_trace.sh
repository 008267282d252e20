# round-0 k_job duration (kernel trace) with every read leaving after phase k (LNR_STOP_AFTER): throughput cost of the phases under
# full concurrency.  (With a stop every read goes to the re-map round afterwards; only the FIRST k_job / k_job_mid of the step is read.)
OUT=gpurun_out/${1:-sat}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
for k in ${STOPS:-2 3 13 4 5 6 7 10 11 8 9 0}; do
  rm -rf $OUT/prof
  LNR_STOP_AFTER=$k timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 tools/grch38_probe.py --batches 1 --reads ${READS:-100000} > $OUT/p_$k.log 2>&1 || { echo "k=$k failed"; tail -3 $OUT/p_$k.log; exit 1; }
  python3 - $OUT $k <<'PY'
import csv, glob, sys
out, k = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(out + "/prof/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("lnr::")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("lnr::k_prep"))
d = {}
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0][5:]
    if n.startswith("k_job_mid"): n = "k_job_mid"
    if n not in d: d[n] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("stop_after %2s: k_job %.2f ms  k_job_mid %.2f ms  seed %.2f" % (k, d.get("k_job", 0), d.get("k_job_mid", 0), d.get("k_seed_fused", 0)))
PY
done
rm -rf $OUT/prof
