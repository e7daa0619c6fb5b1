// k_fwd_bwd instantiations for hidden_units = 256 (num_heads 8 -> 32 channels per head)
#include "tlsan_attn_inst.h"
hipError_t tlsan_launch_fwd_bwd_d256(bool train, bool lstream, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  return launch_fwd_bwd_impl<256, 32>(train, lstream, a, grid, st, ev);
}
