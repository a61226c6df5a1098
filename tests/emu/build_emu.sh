#!/bin/sh
# TEST INFRASTRUCTURE ONLY: compile the product sources against the CPU stand-in for the HIP runtime.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../../signalsmith-stretch_amd/csrc
g++ -O2 -std=c++17 -fPIC -shared -I"$HERE" -x c++ "$SRC/smst_kernels.hip" "$SRC/smst_engine.cpp" "$SRC/smst_capi.cpp" "$HERE/hip_emu.cpp" \
    -o "$HERE/libsmst_emu.so" -Wno-unused-value
echo "built tests/emu/libsmst_emu.so"
