// gemm_bf16.hip - the GEMM / conv kernels of gemm_impl.h instantiated for bf16_t
#include "gemm_impl.h"

template int gemm_run<bf16_t>(const emo_gemm_params&, const GemmPlan&, int, hipStream_t);
