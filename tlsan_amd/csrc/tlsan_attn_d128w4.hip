// k_fwd_bwd for hidden_units = 128 as 4-wavefront workgroups of 8 samples (Geo<128, 16, 4>): training steps with the
// window in registers, no dropout (a translation unit of its own so that it compiles beside the others)
#include "tlsan_attn_inst.h"
hipError_t tlsan_launch_fwd_bwd_d128w4(const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  return launch_train_nw<128, 16, 4>(a, grid, st, ev);
}
