// k_fwd_bwd2 instantiations (one sample per wavefront; d = 64 and 128)
#include "tlsan_attn2.h"
hipError_t tlsan_launch_fwd_bwd2(int D, const FwdArgs& a, int grid, hipStream_t st) {
  if (D == 64) return launch_fwd_bwd2<64, 8>(a, grid, st);
  if (D == 128) return launch_fwd_bwd2<128, 16>(a, grid, st);
  return hipErrorInvalidValue;
}
