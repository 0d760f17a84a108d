// denoiser_fast.hip - MFMA kernels for the benchmark geometry (placeholder: generic dispatch).
#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

bool fast_path_supported(const diffab_dims*) { return false; }
size_t ipa_fast_workspace_floats(const diffab_dims*) { return 0; }
int ipa_layer_fast(const diffab_dims*, const diffab_ipa_layer_weights*, const float*, const float*, const float*, const float*, float*,
                   float*, hipStream_t) {
  set_error("fast path not built");
  return DIFFAB_ERR_UNSUPPORTED;
}
int launch_linear(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int Kd, bool relu,
                  hipStream_t st) {
  return launch_linear_generic(X, ldx, W, bias, Y, ldy, M, N, Kd, relu, st);
}

}  // namespace diffab
