// Backward pass of the hot path (placeholder until the hand-written gradient kernels land).
#include "common.h"

using namespace mpnhip;

extern "C" size_t mpnhip_backward_workspace_bytes(const mpnhip_model*, int, int64_t) { return 0; }

extern "C" int mpnhip_backward(const mpnhip_model*, const void*, int, int64_t, const float*, const float*, const float*,
                               const float*, const float*, float*, float*, void*, size_t, void*, size_t, void*) {
    set_error("mpnhip_backward: not built yet");
    return MPNHIP_ERR_UNSUPPORTED;
}
