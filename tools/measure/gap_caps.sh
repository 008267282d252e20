# bash tools/measure/gap_caps.sh  (on the GPU box): k_gap stage time against the work budget of the lane-per-read launch and the shape
# of the first launch, chr22 stand-in, 20 k reads, -g 50
set -e
cd "$(dirname "$0")/../.."
out=gpurun_out/gap_caps.txt; : > $out
run() {  # name, mode, cap, arena2
  LNR_GAP_MODE=$2 LNR_GAP_WORK_CAP=$3 LNR_GAP_ARENA2_MB=$4 timeout -k 10 300 python bench.py --workload chr22 --reads 20000 --steps 2 --warmup 1 --gap 50 --no-cpu-baseline > gpurun_out/gm.json 2> gpurun_out/gm.log
  python - "$1" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/gm.json"))
s = d["config"]["stage_ms_per_step"]
print(f"{sys.argv[1]:34s} gap {s['gap']:9.1f} ms/step   second-pass reads/step {d['config'].get('gap_second_pass_per_step', -1):8.1f}   {d['value']:.0f} reads/s", flush=True)
PY
  tail -1 $out
}
run "lane cap 3M   arena2 8MB"   0 3000000 8
run "lane cap 300k arena2 8MB"   0 300000 8
run "lane cap 30k  arena2 8MB"   0 30000 8
run "lane cap 3k   arena2 8MB"   0 3000 8
run "lane cap 0    arena2 8MB"   0 0 8
run "wave cap 3M   arena2 8MB"   1 3000000 8
run "wave cap 30k  arena2 8MB"   1 30000 8
run "lane cap 30k  arena2 2MB"   0 30000 2
