# Seed-stage A/B over environment settings on ONE box: bash tools/measure/seed_env_ab.sh name=ENV=VAL[,ENV=VAL] ...  ("name=" alone: no setting)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/seed_ab
for rep in 1 2; do for kv in "$@"; do
  v=${kv%%=*}; rest=${kv#*=}; envs=$(echo $rest | tr ',' ' ')
  env $envs timeout -k 10 200 python tools/grch38_probe.py --seed-only --batches 2 ${PROBE_ARGS} > gpurun_out/seed_ab/$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/seed_ab/$v.log; exit 1; }
  echo "$v: $(grep 'rep 1' gpurun_out/seed_ab/$v.log | sed 's/.*ms prep [0-9.]* seed \([0-9.]*\).*/\1/' | tr '\n' ' ')"
done; done
