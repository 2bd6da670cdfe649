#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG ..."   -> scratch/variants/NAME.so (gemm.hip rebuilt with the flags, other objects from lib/)
set -e
cd "$(dirname "$0")/.."
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $@ -c emote_hack_amd/csrc/gemm.hip -o /tmp/emo_variant_$N.o
L=emote_hack_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/elementwise.o $L/norm.o /tmp/emo_variant_$N.o $L/attention.o $L/temporal.o $L/conditioning.o -o emote_hack_amd/lib/variants/$N.so
echo built emote_hack_amd/lib/variants/$N.so
