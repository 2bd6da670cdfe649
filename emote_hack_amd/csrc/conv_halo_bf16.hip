// conv_halo_bf16.hip - the halo-reuse 3x3 conv of conv_halo_impl.h instantiated for bf16_t
#include "conv_halo_impl.h"

template int gemm_run_halo<bf16_t>(const emo_gemm_params&, int, int, int64_t, hipStream_t);
