#!/bin/sh
# TEST INFRASTRUCTURE ONLY: compile the product sources against the CPU stand-in for the HIP runtime.
# usage: build_emu.sh          -> tests/emu/libsmst_emu.so (what the CPU tests load)
#        build_emu.sh asan     -> tests/emu/libsmst_emu_asan.so: the same sources with -fsanitize=address,undefined; run the CPU suite
#                                 on it with
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
#   SMST_EMU_LIBRARY=tests/emu/libsmst_emu_asan.so python -m pytest tests/test_parity_emu.py tests/test_abi.py -x -q
#                                 (every global / LDS access of every kernel is then bounds-checked: round 3 found an out-of-bounds
#                                 table read of the staged producers this way)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../../signalsmith-stretch_amd/csrc
KERNELS="$SRC/smst_fft.hip $SRC/smst_feed.hip $SRC/smst_vocoder.hip $SRC/smst_vocoder_cont.hip $SRC/smst_vocoder_n.hip $SRC/smst_state.hip"
if [ "$1" = "asan" ]; then
  g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -I"$HERE" -x c++ $KERNELS "$SRC/smst_engine.cpp" \
      "$SRC/smst_capi.cpp" "$HERE/hip_emu.cpp" -o "$HERE/libsmst_emu_asan.so" -Wno-unused-value
  echo "built tests/emu/libsmst_emu_asan.so"
  exit 0
fi
g++ -O2 -std=c++17 -fPIC -shared -I"$HERE" -x c++ $KERNELS "$SRC/smst_engine.cpp" "$SRC/smst_capi.cpp" "$HERE/hip_emu.cpp" \
    -o "$HERE/libsmst_emu.so" -Wno-unused-value
echo "built tests/emu/libsmst_emu.so"
