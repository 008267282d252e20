# kernel trace of the last batch of the GRCh38 probe: start / duration of every lnr kernel (env passes through, e.g. LNR_POST_SPLIT)
OUT=gpurun_out/${1:-ptrace}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT && rm -rf $OUT/prof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 tools/grch38_probe.py --batches 1 ${PROBE_ARGS} > $OUT/probe.log 2>&1 || { tail -5 $OUT/probe.log; exit 1; }
grep "rep 1" $OUT/probe.log
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = sorted(glob.glob(out + "/prof/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("lnr::")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("lnr::k_prep"))
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%-20s start %8.3f ms  dur %8.3f ms  end %8.3f  grid %s" % (r["Kernel_Name"].split("(")[0][5:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
rm -rf $OUT/prof
