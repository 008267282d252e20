# Full-path A/B of several builds / environments on ONE box: bash tools/measure/full_ab.sh name=lib.so[,ENV=VAL...] ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full_ab
for rep in 1 2; do for kv in "$@"; do
  v=${kv%%=*}; rest=${kv#*=}; L=${rest%%,*}; envs=""; [ "$rest" != "$L" ] && envs=$(echo ${rest#*,} | tr ',' ' ')
  env $envs LNR_PROBE_LIB=$L timeout -k 10 200 python tools/grch38_probe.py --batches 2 ${PROBE_ARGS} > gpurun_out/full_ab/$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/full_ab/$v.log; exit 1; }
  echo "$v: $(grep 'rep 1' gpurun_out/full_ab/$v.log | sed 's/rep 1 batch [0-9]: \([0-9.]*\) ms.*job \([0-9.]*\) tail.*/step \1 job \2;/' | tr '\n' ' ')"
done; done
