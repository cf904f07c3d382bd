// k_conv_fused, exact-f32 instantiations of the l <= 1 tensor product (FasterTensorProduct structure, models/tensor_layers.py:71-122):
// static chain shapes with sparse (MODE 0), dense (3) and shared-node (4) rows, 4 or 5 column blocks, and the predicated generic
// variant (1).  Device code: k_conv_tile.h.
#include "k_conv_tile.h"

namespace ddmi {
#define FC_INST(MAXD, SHD, MODE, NBK, BF) template void launch_conv_fused_k<MAXD, SHD, MODE, NBK, BF>(const FusedConvArgs&, hipStream_t);
FC_INST(3, 4, 1, 4, false) FC_INST(3, 4, 0, 4, false) FC_INST(3, 4, 3, 4, false) FC_INST(3, 4, 4, 4, false)
FC_INST(3, 4, 0, 5, false) FC_INST(3, 4, 3, 5, false) FC_INST(3, 4, 4, 5, false)
#undef FC_INST
#ifdef DDMI_PROFILING
void fc_prof_report_f32() {
  fc_wg_dump();
#ifdef DDMI_PHASE_CLOCKS
  fc_prof_report_tu();
#endif
}
#endif
}  // namespace ddmi
