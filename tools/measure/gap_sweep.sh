# gap stage per step for a few (teams, heavy weight) settings of the fused launch: bench.py --gap 50 without the CPU legs
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3x
for cfg in "96 60000" "64 100000" "64 150000" "48 200000" "96 120000" "128 40000" "80 80000"; do
  set -- $cfg
  LNR_GAP_TEAMS=$1 LNR_GAP_HEAVY_W=$2 timeout -k 10 200 python3 bench.py --gap 50 --steps 4 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/r3x/g_$1_$2.json 2> gpurun_out/r3x/g_$1_$2.log || { echo "failed $cfg"; tail -3 gpurun_out/r3x/g_$1_$2.log; exit 1; }
  python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r3x/g_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
c = d["config"]
print(f"teams {sys.argv[1]:>4} heavy_w {sys.argv[2]:>7}: gap stage {c['stage_ms_per_step']['gap']:.1f} ms, device-resident {c['device_resident_reads_per_s']:.0f} reads/s, team reads per step {c.get('gap_second_pass_per_step')}", flush=True)
PY
done
