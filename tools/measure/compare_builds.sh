# Kernel-trace bench of several builds of the library, twice each, in one gpurun call (box-to-box spread is larger than most
# effects, so A and B must run on the same box):  bash tools/measure/compare_builds.sh base=path/lib_base.so new=linear_amd/liblinear_amd.so
# Prints ms per step and the durations of the job / prep / seed / tail kernels of the last steps.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
for rep in 1 2; do for kv in "$@"; do
  v=${kv%%=*}; L=${kv#*=}
  rm -rf gpurun_out/ab/t_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab/t_$v -- python3 tools/bench_variant.py $L --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ab/b_$v.json 2> gpurun_out/ab/l_$v.log || exit 1
  python3 - <<PY
import csv,glob,json,collections
f=sorted(glob.glob("gpurun_out/ab/t_$v/*/*kernel_trace.csv"))[-1]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if n.startswith("lnr::k_job") or n.startswith("lnr::k_prep") or n.startswith("lnr::k_seed") or n.startswith("lnr::k_tail") or n.startswith("lnr::k_f1"): d[n.split("(")[0][5:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
b=json.loads(open("gpurun_out/ab/b_$v.json").read().strip().splitlines()[-1])
print("%-6s %.2f ms " % ("$v", b["ms_per_step"]), {k:[round(x,2) for x in v[-4:]] for k,v in sorted(d.items())})
PY
done; done
