# kernel trace of the default bench (GRCh38 stand-in) + in-kernel phase stamps of the job kernels (-DLNR_PROF build)
OUT=gpurun_out/${1:-r2prof}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT && rm -rf $OUT/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log || { tail -5 $OUT/bench_under_rocprof.log; exit 1; }
cp $(ls $OUT/prof/*/*kernel_stats.csv | tail -1) $OUT/kernel_stats.csv
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = sorted(glob.glob(out + "/prof/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("lnr::")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last k_prep on
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("lnr::k_prep"))
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%-28s start %8.3f ms  dur %8.3f ms  grid %s wg %s" % (r["Kernel_Name"].split("(")[0][5:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
PY
head -14 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof
timeout -k 10 300 python3 tools/prof_job_phases.py 100000 grch38 > $OUT/phases.log 2>&1; grep -E "^==|^   [a-z]|biggest" $OUT/phases.log | cut -c1-260
