// conv_halo_f32.hip - the halo-reuse 3x3 conv of conv_halo_impl.h instantiated for float
#include "conv_halo_impl.h"

template int gemm_run_halo<float>(const emo_gemm_params&, int, int, int64_t, hipStream_t);
