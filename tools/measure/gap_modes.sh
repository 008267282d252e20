# bash tools/measure/gap_modes.sh  (on the GPU box): the gap re-mapper's stage time (k_gap, -g 50) for the worker shapes and
# register budgets of the kernel -- lane per read / wave per read, 1 / 4 / 8 waves per SIMD -- on the chr22 stand-in, 20 k reads.
set -e
cd "$(dirname "$0")/../.."
out=gpurun_out/gap_modes.txt; : > $out
run() {  # name, lib, mode, waves
  LNR_LIB=$2 LNR_GAP_MODE=$3 LNR_GAP_WAVES=$4 timeout -k 10 300 python bench.py --workload chr22 --reads 20000 --steps 2 --warmup 1 --gap 50 --no-cpu-baseline > gpurun_out/gm.json 2> gpurun_out/gm.log
  python - "$1" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/gm.json"))
s = d["config"]["stage_ms_per_step"]
print(f"{sys.argv[1]:28s} gap {s['gap']:9.1f} ms/step   total {d['ms_per_step']:9.1f} ms/step   {d['value']:.0f} reads/s", flush=True)
PY
  tail -1 $out
}
run "lane/read  occ1"  ""                         0 0
run "wave/read  occ1"  ""                         1 16384
run "lane/read  occ4"  tools/_variants/gapw4.so   0 0
run "wave/read  occ4"  tools/_variants/gapw4.so   1 16384
run "wave/read  occ8"  tools/_variants/gapw8.so   1 16384
run "lane/read  occ8"  tools/_variants/gapw8.so   0 0
