#!/bin/bash
# Probe builds of the library for tools/interfere_probe.py: the in-tree objects with conv_x3.o rebuilt under a switch.
#   tools/_alt/libmliis_noclaim.so    -DX3_NO_CLAIM                conv_x3_k with its natural register count (co-resident with other kernels' waves)
#   tools/_alt/libmliis_neither.so    -DX3_NO_CLAIM -DF3_NO_CLAIM  neither split-product kernel claims its CU
#   tools/_alt/libmliis_fnoclaim.so   -DF3_NO_CLAIM                the round-5 arrangement (conv_x3_k claims, conv_filter_x3_batched_k does not)
set -e
cd "$(dirname "$0")/../mliis_amd/csrc"
make -j8 >/dev/null
OUT=../../tools/_alt
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize"
SRC=${SRC:-conv_x3}       # the translation unit rebuilt under the switch (SRC=dwmarch tools/build_variants.sh "ts2:-DDWM_K5_TS2=1")
[ $SRC = conv_x3 ] || FLAGS=${FLAGS/ -fno-slp-vectorize/}
OBJS=$(ls *.o | grep -v $SRC.o)
# further variants: tools/build_variants.sh "name:-Dflag ..." ...
if [ $# -gt 0 ]; then VARIANTS=("$@"); else VARIANTS=("noclaim:-DX3_NO_CLAIM" "neither:-DX3_NO_CLAIM -DF3_NO_CLAIM" "fnoclaim:-DF3_NO_CLAIM"); fi
for v in "${VARIANTS[@]}" ; do
  name=${v%%:*}; def=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $def -c $SRC.hip -o $OUT/${SRC}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o $OUT/libmliis_$name.so $OBJS $OUT/${SRC}_$name.o
done
ls -la $OUT
