// gemm_f32.hip - the GEMM / conv kernels of gemm_impl.h instantiated for float
#include "gemm_impl.h"

template int gemm_run<float>(const emo_gemm_params&, const GemmPlan&, int, hipStream_t);
template int gemm_run_halo<float>(const emo_gemm_params&, int, int, int64_t, hipStream_t);
