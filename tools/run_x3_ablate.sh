#!/bin/bash
# Ablations of conv_x3_k (probe builds: wrong numbers by design): launch time of the 224 -> 112 conv at 56x56, forward.
cd $GRAFT_REPO_ROOT
for V in base $VARIANTS; do
  if [ $V = base ]; then unset MLIIS_HIP_LIB; else export MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_$V.so; fi
  printf "%-10s " $V; python tools/x3_probe.py 40 rsd2.fuse 2>/dev/null | grep " fwd " | sed 's/.*x3k err/x3k err/'
done
