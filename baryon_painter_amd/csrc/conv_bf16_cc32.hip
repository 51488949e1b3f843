// igemm_bf16_kernel / igemm_bf16_p_kernel instantiated for channel chunks of 32 (see conv_bf16.hpp).
#include "conv_bf16.hpp"

int bp_bf16_launch_cc32(const bpbf16::BConfig& c, const bpbf16::BArgs& a, bool in_bf16, bool out_bf16, dim3 grid,
                          hipStream_t st) {
  return bpbf16::b_launch_cc<32>(c, a, in_bf16, out_bf16, grid, st);
}
