// k_fwd_bwd instantiations for hidden_units = 256 (num_heads 8 -> 32 channels per head), windows in registers (Ls <= 10);
// the streamed form is tlsan_attn_d256s.hip
#include "tlsan_attn_inst.h"
hipError_t tlsan_launch_fwd_bwd_d256s(bool train, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev);
hipError_t tlsan_launch_fwd_bwd_d256(bool train, bool lstream, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  if (lstream) return tlsan_launch_fwd_bwd_d256s(train, a, grid, st, ev);
  return launch_fwd_bwd_form<256, 32, false>(train, a, grid, st, ev);
}
