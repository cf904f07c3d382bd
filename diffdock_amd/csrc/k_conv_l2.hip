// k_conv_fused for sh_lmax = 2 (nine harmonics per edge) and second-order node irreps (output blocks up to l = 2): the e3nn
// FullyConnectedTensorProduct layers of models/tensor_layers.py:292-300.  Device code: k_conv_tile.h.
#include "k_conv_tile.h"

namespace ddmi {
#define FC_INST(MAXD, SHD, MODE, NBK, BF) template void launch_conv_fused_k<MAXD, SHD, MODE, NBK, BF>(const FusedConvArgs&, hipStream_t);
FC_INST(3, 9, 1, 4, false) FC_INST(3, 9, 0, 4, false) FC_INST(3, 9, 3, 4, false)
FC_INST(5, 9, 1, 4, false) FC_INST(5, 9, 0, 4, false) FC_INST(5, 9, 3, 4, false)
#undef FC_INST
#ifdef DDMI_PROFILING
void fc_prof_report_l2() {
  fc_wg_dump();
#ifdef DDMI_PHASE_CLOCKS
  fc_prof_report_tu();
#endif
}
#endif
}  // namespace ddmi
