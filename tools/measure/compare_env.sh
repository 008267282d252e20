# Same, for one build under several environments (library tunables are read at lnr_create):
#   bash tools/measure/compare_env.sh "default:LNR_X=1" "lds7:LNR_JOB_LDS_KB=7"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
for kv in "$@"; do
  v=${kv%%:*}; E=${kv#*:}
  rm -rf gpurun_out/ab/e_$v
  env $E timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab/e_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/eb_$v.json 2> gpurun_out/ab/el_$v.log || exit 1
  python3 - <<PY
import csv,glob,json,collections
f=sorted(glob.glob("gpurun_out/ab/e_$v/*/*kernel_trace.csv"))[-1]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if n.startswith("lnr::k_job") or n.startswith("lnr::k_prep") or n.startswith("lnr::k_seed"): d[n.split("(")[0][5:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
b=json.loads(open("gpurun_out/ab/eb_$v.json").read().strip().splitlines()[-1])
print("%-10s %.2f ms " % ("$v", b["ms_per_step"]), {k:[round(x,2) for x in v[-4:]] for k,v in sorted(d.items())})
PY
done
