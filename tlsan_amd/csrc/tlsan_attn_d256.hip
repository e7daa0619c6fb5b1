// k_fwd_bwd instantiations for hidden_units = 256 (num_heads 8 -> 32 channels per head), windows in registers (Ls <= 10);
// the streamed form is tlsan_attn_d256s.hip.  Built with -mllvm -sink-insts-to-avoid-spills (tlsan_amd/build.py): the machine sinking
// pass moves invariant address arithmetic back into the loops that use it instead of keeping it in registers that get
// spilled (scratch loads of the Ls = 10 training kernel 108 -> 69, reloads that drain the vector-memory queue 26 -> 5):
// Ls = 10 174 -> 167 us/step, in bf16 135 -> 124 (profiles/r04_d256_sink_ab.md); the d = 64 / 128 units lose with it.
#include "tlsan_attn_inst.h"
hipError_t tlsan_launch_fwd_bwd_d256s(bool train, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev);
hipError_t tlsan_launch_fwd_bwd_d256(bool train, bool lstream, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  if (lstream) return tlsan_launch_fwd_bwd_d256s(train, a, grid, st, ev);
  return launch_fwd_bwd_form<256, 32, false>(train, a, grid, st, ev);
}
