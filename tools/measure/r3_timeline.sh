# kernel timeline of one step (last k_prep .. end) under rocprofv3 --kernel-trace: bash tools/measure/r3_timeline.sh <outdir> [bench args]
OUT=gpurun_out/${1:-r3tl}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT && rm -rf $OUT/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli "$@" > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log || { tail -5 $OUT/bench_under_rocprof.log; exit 1; }
cp $(ls $OUT/prof/*/*kernel_stats.csv | tail -1) $OUT/kernel_stats.csv
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = sorted(glob.glob(out + "/prof/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("lnr::")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
preps = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("lnr::k_prep")]
idx = preps[-1]
t0 = int(rows[idx]["Start_Timestamp"])
with open(out + "/timeline_last_step.txt", "w") as fo:
    for r in rows[idx:]:
        line = "%-28s start %9.3f ms  dur %9.3f ms  grid %s wg %s" % (r["Kernel_Name"].split("(")[0][5:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
        fo.write(line + "\n")
        if float(line.split("dur")[1].split("ms")[0]) > 0.3:
            print(line)
PY
head -12 $OUT/kernel_stats.csv | cut -c1-140
rm -rf $OUT/prof
