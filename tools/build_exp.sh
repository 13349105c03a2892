#!/bin/bash
# Builds an experimental copy of the library with extra compiler flags for ONE source file:
#   tools/build_exp.sh <name> <source.hip> <flags...>   ->  gpurun_exp/<name>.so
# (load it with MIF_LIBRARY=gpurun_exp/<name>.so; the other objects come from the regular build)
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../object_detection_amd/csrc"
make -s -j8
mkdir -p ../../gpurun_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" \
    -c "$SRC" -o "../../gpurun_exp/$NAME.o"
OBJS=""
for f in build/*.o; do
    [ "$(basename "$f" .o)" = "$(basename "$SRC" .hip)" ] && OBJS="$OBJS ../../gpurun_exp/$NAME.o" || OBJS="$OBJS $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../../gpurun_exp/$NAME.so" $OBJS -ldl
echo "built gpurun_exp/$NAME.so"
