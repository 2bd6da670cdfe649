#!/bin/bash
# usage: build_variant.sh NAME SRC "-DFLAG ..."   -> emote_hack_amd/lib/variants/NAME.so (SRC.hip rebuilt with the flags, other objects from lib/)
set -e
cd "$(dirname "$0")/.."
N=$1; SRC=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $@ -c emote_hack_amd/csrc/$SRC.hip -o /tmp/emo_variant_$N.o
L=emote_hack_amd/lib
OBJS=""
for o in elementwise norm gemm gemm_f32 gemm_bf16 gemm_f16 attention temporal conditioning; do
  if [ "$o" == "$SRC" ]; then OBJS="$OBJS /tmp/emo_variant_$N.o"; else OBJS="$OBJS $L/$o.o"; fi
done
mkdir -p $L/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $L/variants/$N.so
echo built $L/variants/$N.so
