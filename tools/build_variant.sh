#!/bin/bash
# Diagnostic: builds tools/_build/libisg_<name>.so with extra -D flags on ONE source file (the rest from the
# objects of the last regular build).  usage: tools/build_variant.sh <name> <source.hip> [-DFLAG ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
mkdir -p $R/tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function "$@" -c $R/imagesegmentation_amd/csrc/$SRC -o $R/tools/_build/${NAME}.o
OBJS=$(ls $R/imagesegmentation_amd/csrc/_obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_build/libisg_${NAME}.so $OBJS $R/tools/_build/${NAME}.o
echo built $R/tools/_build/libisg_${NAME}.so
