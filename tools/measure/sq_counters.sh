# SQ instruction-mix counters of the job and seed kernels in three rocprofv3 --pmc passes (profiles/r01/v4_pmc_sq_counters.json).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/sq
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_FLAT_LDS_ONLY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1)); rm -rf gpurun_out/sq/p$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/sq/p$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/sq/b$i.json 2> gpurun_out/sq/l$i.log || { tail -3 gpurun_out/sq/l$i.log; continue; }
done
python3 - <<PY
import csv,glob,collections,json
out=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0]
        if n.startswith("lnr::k_job") or n.startswith("lnr::k_seed"):
            out[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={k:{c:v for c,v in d.items()} for k,d in out.items()}
json.dump(res, open("gpurun_out/sq/summary.json","w"), indent=1)
for k,d in res.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-28s %s" % (c, [round(x/1e6,1) for x in v]))
PY
