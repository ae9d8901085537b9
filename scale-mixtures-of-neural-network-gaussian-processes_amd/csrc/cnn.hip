// cnn.hip — conv-NNGP kernel (experiments/nt_kernels.py:34-45).  Placeholder until the pair-tile
// kernel lands: reports SMN_ENOTSUP so callers fail loudly instead of silently falling back.
#include "internal.hpp"

extern "C" int smn_kernel_cnn(smn_ctx* ctx, int dtype, int act, int num_hiddens, double w_std, double b_std,
                              double last_w_std, const void* x1_d, int64_t n1, const void* x2_d, int64_t n2, int64_t H,
                              int64_t W, int64_t C, int fill, void* nngp_d, int64_t ldk) {
  (void)dtype; (void)act; (void)num_hiddens; (void)w_std; (void)b_std; (void)last_w_std; (void)x1_d; (void)n1;
  (void)x2_d; (void)n2; (void)H; (void)W; (void)C; (void)fill; (void)nngp_d; (void)ldk;
  return smn_fail(ctx, SMN_ENOTSUP, "smn_kernel_cnn: not implemented yet");
}
