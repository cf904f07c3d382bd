// Grouped dispatch of the fused convolution (round 6): ONE launch per interaction layer walks the work items (edge group, tile of
// 16 virtual nodes, granule range) of every edge group of the layer.  The groups of a TensorProductConvLayer are independent
// until the joint mean over all incoming messages (models/tensor_layers.py:148-231: tp_scatter_multigroup), so they need no
// order among themselves; per-group launches on two streams (k_conv_fused) leave a launch tail per group, keep small groups
// (lig-lig: 10-79 tiles) from ever filling the chip, and chain hidden rows -> convolution per group.  A workgroup decodes its
// item from blockIdx.x, switches to the body of the group's row mode (fc_tile<.., MODE, ..>: the same device code as the
// per-group kernels, k_conv_tile.h) and runs it unchanged: results are bit-identical to the per-group launches.
#include "k_conv_tile.h"

namespace ddmi {

template <int NBK>
__global__ __launch_bounds__(512) void k_conv_grouped(FusedGroupedArgs G) {
  DDMI_DYN_SMEM(float, smem);
  const int b = (int)blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < FC_GROUPS_MAX; ++i)
    if (i < G.n && b >= G.first[i]) g = i;
  const int loc = b - G.first[g], nt = G.ntile[g];
  const int by = loc / nt, bx = loc - by * nt;
  const FusedConvArgs& a = G.g[g];
  const int mode = G.mode[g];
  if (mode == 4) fc_tile<3, 4, 4, NBK, false>(a, bx, by, smem);
  else if (mode == 3) fc_tile<3, 4, 3, NBK, false>(a, bx, by, smem);
  else fc_tile<3, 4, 0, NBK, false>(a, bx, by, smem);
}

template <int NBK>
static void launch_conv_grouped_k(const FusedGroupedArgs& G, size_t smem, hipStream_t s) {
  static bool lds_opt_in = false;
  if (!lds_opt_in) {
    DDMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_grouped<NBK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_opt_in = true;
  }
  hipLaunchKernelGGL((k_conv_grouped<NBK>), dim3(G.first[G.n]), dim3(64 * FC_WAVES), smem, s, G);
  DDMI_CHECK_HIP(hipGetLastError());
}

void launch_conv_grouped(const FusedGroupedArgs& G_in, hipStream_t s) {
  FusedGroupedArgs G = G_in;
  if (G.n <= 0) return;
  if (G.n > FC_GROUPS_MAX) throw Error(DDMI_ERR_ARG, "k_conv_grouped: too many edge groups in one launch");
  int max_nb = 4, max_local = 0, blocks = 0;
  for (int i = 0; i < G.n; ++i) {
    FusedConvArgs& a = G.g[i];
    a.dbg = ablate_mask();
    if (a.generic || a.bf || a.maxd > 3 || a.sh_lmax > 1)
      throw Error(DDMI_ERR_ARG, "k_conv_grouped: exact-f32 l <= 1 layers with static chain shapes only (complex.cpp routes the others per group)");
    G.mode[i] = (a.dense && a.shared) ? 4 : a.dense ? 3 : 0;
    // (the pre-reduction contract of launch_conv_fused: only the mode 0 / 3 bodies honour tile_hdr)
    if (a.tile_hdr && G.mode[i] == 4) throw Error(DDMI_ERR_STATE, "k_conv_grouped: pre-reduced group routed to the shared-node body");
    max_nb = std::max(max_nb, a.max_nb);
    max_local = std::max(max_local, fc_max_local(a));
    G.first[i] = blocks;
    G.ntile[i] = std::max(1, cdiv(a.vcap, FC_VN));
    blocks += (a.vcap > 0 ? G.ntile[i] * a.ysplit : 0);
  }
  G.first[G.n] = blocks;
  if (blocks <= 0) return;
  const size_t smem = max_nb > 4 ? fc_smem_bytes<3, 4, 3, 5>(max_local) : fc_smem_bytes<3, 4, 3, 4>(max_local);
  if (smem > 160 * 1024) throw Error(DDMI_ERR_CAPACITY, "k_conv_grouped: LDS budget exceeded (raise ddmi_config.exec.tile_split)");
  if (max_nb > 4) launch_conv_grouped_k<5>(G, smem, s);
  else launch_conv_grouped_k<4>(G, smem, s);
}

#ifdef DDMI_PROFILING
void fc_prof_report_grp() {
  fc_wg_dump();
#ifdef DDMI_PHASE_CLOCKS
  fc_prof_report_tu();
#endif
}
#endif
}  // namespace ddmi
