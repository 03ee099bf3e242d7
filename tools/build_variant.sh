#!/bin/bash
# Build a variant of the library that differs in ONE source:  bash tools/build_variant.sh <name> <source.hip> "<extra flags>"
# -> ct-clip-ut_amd/ctclip_hip/libctclip_hip_<name>.so (the other objects come from the product build directory)
set -e
NAME=$1; SRC=$2; EXTRA=$3
PKG=$(dirname $0)/../ct-clip-ut_amd
mkdir -p $PKG/ctclip_hip/_build_var
OBJ=$PKG/ctclip_hip/_build_var/${NAME}_$(basename $SRC .hip).o
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -fPIC -std=c++17 -Wno-unused-value $EXTRA -c $PKG/csrc/$SRC -o $OBJ
OTHERS=$(ls $PKG/ctclip_hip/_build/*.o | grep -v "/$(basename $SRC .hip).o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/ctclip_hip/libctclip_hip_$NAME.so $OBJ $OTHERS
echo built libctclip_hip_$NAME.so
