#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG ..."   -> emote_hack_amd/lib/variants/NAME.so
#        TUS="attention" build_variant.sh att_qt1 -DEMO_ATT_QT=1     (translation units that see the flags; default: gemm gemm_bf16)
#        PATCH=tools/bench/patches/gemm_timing.patch build_variant.sh timing   (sources copied to /tmp and patched first)
# Rebuilds the GEMM translation units that see the flags (gemm.hip: planning, gemm_bf16.hip: kernels) and links them with the
# other objects of the product build.  BENCH ONLY: the f32 / f16 kernels keep the product's geometry - load the variant with
# EMO_HIP_LIB=... and run bf16 microbenchmarks (tools/bench/gemm_tiles.py).
set -e
cd "$(dirname "$0")/../.."
N=$1; shift
SRC=emote_hack_amd/csrc
if [ -n "$PATCH" ]; then
  T=/tmp/emo_variant_src_$N; rm -rf $T; mkdir -p $T/x; cp -r $SRC $T/x/csrc; cp -r include $T/include   # (common.h: ../../include/emo_hip.h)
  patch -s -p2 -d $T/x/csrc < $PATCH
  SRC=$T/x/csrc
fi
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed"
TUS=${TUS:-"gemm gemm_bf16"}
for tu in $TUS; do
  /opt/rocm/bin/hipcc $FL $@ -c $SRC/$tu.hip -o /tmp/emo_variant_${N}_$tu.o &
done
wait
L=emote_hack_amd/lib
OBJS=""
for o in elementwise norm gemm gemm_f32 gemm_bf16 gemm_f16 conv_halo_f32 conv_halo_bf16 conv_halo_f16 attention temporal conditioning frontend; do
  if [[ " $TUS " == *" $o "* ]]; then OBJS="$OBJS /tmp/emo_variant_${N}_$o.o"; else OBJS="$OBJS $L/$o.o"; fi
done
mkdir -p $L/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $L/variants/$N.so
python3 -c "import ctypes,os; ctypes.CDLL(os.path.abspath('$L/variants/$N.so'), mode=os.RTLD_NOW)"   # every kernel stub resolves
echo built $L/variants/$N.so
