#!/usr/bin/env bash
# Builds differently-flagged variants of libbgamd.so from the CURRENT sources for A/B measurements on one GPU box:
#   bash tools/ab_build.sh name1 "-DFLAG=1 -DOTHER=2" name2 "" ...
# -> backgammon-engine_amd/variants/libbgamd_<name>.so (git-ignored, travels with gpurun); run them with tools/ab_run.py
set -e
cd "$(dirname "$0")/.."
H=$(python3 -c "import sys; sys.path.insert(0,'backgammon-engine_amd/backgammon_env'); import _srchash; print(_srchash.source_hash())")
mkdir -p backgammon-engine_amd/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "-DBGAMD_SRC_HASH=\"$H\"" $flags \
      -Xclang -target-feature -Xclang -bitop3-insts backgammon-engine_amd/csrc/bgamd.hip -o backgammon-engine_amd/variants/libbgamd_$name.so 2>&1 | grep -v "bitop3-insts" || true
  echo "built $name [$flags]"
done
