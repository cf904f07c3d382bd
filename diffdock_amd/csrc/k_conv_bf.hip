// k_conv_fused with the split-bf16 edge product (ddmi_config.edge_product = 1; secondary bench line): the static l <= 1 loops of
// k_conv_f32.hip with hidden rows and contracted chunks as packed bf16 hi | lo words on v_mfma_f32_16x16x32_bf16.  Device code: k_conv_tile.h.
#include "k_conv_tile.h"

namespace ddmi {
#define FC_INST(MAXD, SHD, MODE, NBK, BF) template void launch_conv_fused_k<MAXD, SHD, MODE, NBK, BF>(const FusedConvArgs&, hipStream_t);
FC_INST(3, 4, 0, 4, true) FC_INST(3, 4, 3, 4, true) FC_INST(3, 4, 4, 4, true)
FC_INST(3, 4, 0, 5, true) FC_INST(3, 4, 3, 5, true) FC_INST(3, 4, 4, 5, true)
#undef FC_INST
#ifdef DDMI_PROFILING
void fc_prof_report_bf() {
  fc_wg_dump();
#ifdef DDMI_PHASE_CLOCKS
  fc_prof_report_tu();
#endif
}
#endif
}  // namespace ddmi
