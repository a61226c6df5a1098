#!/bin/bash
# Builds a VARIANT of the product library from a patched copy of csrc/ into signalsmith-stretch_amd/variants/<name>.so
# (git-ignored; travels to the GPU box).  The product sources are never touched.
# usage: tools/probes/build_variant.sh <name> [patch-script.py ...] [-- extra hipcc flags]     e.g.  build_variant.sh trace tools/probes/voc_trace_patch.py
#        tools/probes/build_variant.sh base --git <rev>          (the library of another commit, for same-box A/B runs)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
TMP=$(mktemp -d /tmp/smst_variant.XXXXXX)
mkdir -p $TMP/signalsmith-stretch_amd/csrc $TMP/include $ROOT/signalsmith-stretch_amd/variants
if [ "$1" = "--git" ]; then
  git -C $ROOT archive $2 signalsmith-stretch_amd/csrc include | tar -x -C $TMP
  shift 2
else
  cp $ROOT/signalsmith-stretch_amd/csrc/*.{h,hip,cpp} $TMP/signalsmith-stretch_amd/csrc/
  cp -r $ROOT/include/* $TMP/include/
fi
# the patch scripts address ONE kernel file: the variant is built from the translation units joined into smst_kernels.hip
# (the headers are `#pragma once`, every unit opens and closes the namespace itself); a --git revision that still has the single file is used as it is
K=$TMP/signalsmith-stretch_amd/csrc
if [ ! -f $K/smst_kernels.hip ]; then
  for unit in smst_fft smst_feed smst_vocoder smst_vocoder_cont smst_vocoder_n smst_state; do  # (a --git revision may predate a unit)
    if [ -f $K/$unit.hip ]; then cat $K/$unit.hip >> $K/smst_kernels.hip; fi
  done
fi
EXTRA=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; EXTRA="$*"; break; fi
  python3 "$1" $TMP/signalsmith-stretch_amd/csrc
  shift
done
C=$TMP/signalsmith-stretch_amd/csrc
hipcc --offload-arch=gfx950 -I$C -O3 -ffp-contract=on -std=c++17 -fPIC -shared -x hip -Wno-unused-result -Wno-unused-value $EXTRA \
  $C/smst_kernels.hip $C/smst_engine.cpp $C/smst_capi.cpp -o $ROOT/signalsmith-stretch_amd/variants/$NAME.so 2>&1 | grep -E "error" || true
ls -la --time-style=full-iso $ROOT/signalsmith-stretch_amd/variants/$NAME.so
rm -rf $TMP
