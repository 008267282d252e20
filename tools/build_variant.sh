# bash tools/build_variant.sh <name> [-DFLAG ...]  ->  tools/_variants/<name>.so : the library built from the working tree with extra defines (A/B and diagnostic builds)
set -e
cd "$(dirname "$0")/.."
python -m linear_amd.build "$@"
