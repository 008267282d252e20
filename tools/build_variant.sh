# bash tools/build_variant.sh <name> [-DFLAG ...]  ->  tools/_variants/<name>.so : the library built from the working tree with extra defines (A/B and diagnostic builds)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$@" -o tools/_variants/$name.so linear_amd/csrc/lnr_api.hip \
  -Wl,linear_amd/lnr_reader.o -Wl,linear_amd/lnr_output.o -lz -lpthread
echo tools/_variants/$name.so
