// bf16 MFMA GEMM (fast path).  Placeholder until the tuned kernel lands: reports "unsupported" so
// every call is served by the exact-f32 kernel.
#include "rmcl_common.h"
#include "kernels.h"
bool rmcl_gemm_fast_supported(const GemmArgs&, int, int, int, int) { return false; }
int rmcl_launch_gemm_fast(const GemmArgs&, int, int, int, hipStream_t) { rmcl_set_error("fast gemm not built"); return -1; }
