// gemm_f32.hip - the GEMM / conv kernels of gemm_impl.h instantiated for float
#include "gemm_impl.h"

template int gemm_run<float>(const emo_gemm_params&, const GemmPlan&, int, hipStream_t);
