// pending_stubs.hip -- entry points whose kernels have not landed yet.  They fail loudly
// (MLPL_E_UNSUPPORTED); nothing here computes on the host.  Entries move out of this file as they land.
#include "mlpl_internal.h"

namespace mlpl {
int launch_knn_l2_mfma(mlpl_ctx *, const float *, int, size_t, size_t, const float *, int, size_t, size_t, int, int, int,
                       int32_t *, float *, hipStream_t, int force) {
    if (force) {
        set_error("fp16 MFMA L2 path not built yet");
        return MLPL_E_UNSUPPORTED;
    }
    return 1;
}
}  // namespace mlpl

