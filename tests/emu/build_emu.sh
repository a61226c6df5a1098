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
# one compiler process per translation unit, side by side (a single g++ over all of them took 74 s of the CPU suite's time)
build() { # build <output> <object dir> <flags...>
  out=$1; obj=$2; shift 2
  mkdir -p "$obj"
  pids=""
  for f in $KERNELS "$SRC/smst_engine.cpp" "$SRC/smst_capi.cpp" "$HERE/hip_emu.cpp"; do
    g++ "$@" -std=c++17 -fPIC -I"$HERE" -x c++ -c "$f" -o "$obj/$(basename "$f").o" -Wno-unused-value &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  g++ "$@" -shared "$obj"/*.o -o "$out"
}
if [ "$1" = "asan" ]; then
  build "$HERE/libsmst_emu_asan.so" "$HERE/obj_asan" -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer
  echo "built tests/emu/libsmst_emu_asan.so"
  exit 0
fi
build "$HERE/libsmst_emu.so" "$HERE/obj" -O2
echo "built tests/emu/libsmst_emu.so"
