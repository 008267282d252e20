# Instruction-cache and wait counters of the job kernels (one rocprofv3 --pmc pass per set; no trace domains beside --pmc).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ic
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_FLAT"; do
  i=$((i+1)); rm -rf gpurun_out/ic/p$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/ic/p$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-gap50 > gpurun_out/ic/b$i.json 2> gpurun_out/ic/l$i.log || { tail -3 gpurun_out/ic/l$i.log; continue; }
done
python3 - <<PY
import csv,glob,collections
out=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/ic/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0]
        if n.startswith("lnr::k_job") or n.startswith("lnr::k_seed"):
            out[n][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,d in out.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-22s %.1f M" % (c, v/1e6))
PY
