#!/bin/bash
# per-kernel register / scratch / LDS usage of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|Occupancy|LDS Size" | sed 's/^[^ ]* remark: *//; s/ *\[-Rpass.*//' |
  awk '/Function Name/{if(n)print n,v,a,s,sp,o,l; n=$3} /^VGPRs:/{v="vgpr="$2} /^AGPRs:/{a="agpr="$2} /ScratchSize/{s="scratch="$4} /VGPRs Spill/{sp="spill="$3} /Occupancy/{o="occ="$4} /LDS Size/{l="lds="$5} END{print n,v,a,s,sp,o,l}' | c++filt
