// k_fwd_bwd instantiations for hidden_units = 256, streamed windows (the list form).  Built with
// -mllvm -disable-machine-licm (tlsan_amd/build.py): the machine-level hoisting of loop invariants out of the pass loop
// parks per-lane addresses in registers that get spilled, and every reload drains the vector-memory queue
// (DESIGN.md 7).  Without it: Ls = 90 280.6 -> 274.7 us/step, C5 in bf16 259.7 -> 249.6; the window-in-registers forms of
// every width lose 1-2 % with the same flag, so they keep the default.  And with -mllvm -sink-insts-to-avoid-spills, as
// tlsan_attn_d256.hip: Ls = 90 275 -> 266 us/step, C5 344 -> 326, C5 in bf16 251 -> 237 (no scratch memory at all there).
#include "tlsan_attn_inst.h"
hipError_t tlsan_launch_fwd_bwd_d256s(bool train, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  return launch_fwd_bwd_form<256, 32, true>(train, a, grid, st, ev);
}
