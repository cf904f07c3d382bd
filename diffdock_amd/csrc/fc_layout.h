// Fragment-order layout of the hidden rows shared by their producer (k_hidden.hip) and their consumer (k_conv.hip).
#pragma once
#include "ddmi_common.h"

namespace ddmi {

// Hidden rows of the edge MLP in virtual-node order and in the A-fragment order of k_conv_fused, two 8-k groups per float4:
//   Hb[v][rt][g >> 1][lane = 16q + r][2 (g & 1) + sub] = relu(HE[arow] + P[tgt] + Q[d])[k = 8g + 2q + sub]   (edge row el = 16rt + r)
// (zero for k >= H and for the padding rows el >= ne): a wave fetches one (row tile, PAIR of 8-k groups) as 1 KB contiguous,
// and the MFMA first layer (k_edge_hidden_mm) writes it with one float4 per lane.
__host__ __device__ __forceinline__ int fc_ngp(int NG8) { return (NG8 + 1) >> 1; }
__device__ __forceinline__ size_t fc_hb_off(int v, int rt, int g, int lane, int NGP) {
  return ((((size_t)v * 2 + rt) * NGP + (g >> 1)) * 64 + lane) * 4 + 2 * (g & 1);
}

}  // namespace ddmi
