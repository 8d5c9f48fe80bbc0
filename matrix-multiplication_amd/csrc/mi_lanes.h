// Device-side lane helpers shared by the SpMM translation units (not installed).
#ifndef MI_LANES_H_
#define MI_LANES_H_

#include <hip/hip_runtime.h>

#include <type_traits>

namespace mi {

template <int N_, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N_ > 0) {
    static_for<N_ - 1>(f);
    f(std::integral_constant<int, N_ - 1>{});
  }
}

// Entries a G-lane group keeps per chunk register.  With SPLIT32 a 32-lane group keeps the SAME 16 entries in both
// of its 16-lane DPP rows (lane l holds entry l & 15) so that the row broadcast reaches it; every other width (and a
// 32-lane group without SPLIT32) keeps one entry per lane.
template <int G, bool SPLIT32>
struct LaneChunk {
  static constexpr int ENTRIES = (G == 32 && SPLIT32) ? 16 : G;
};

#ifndef MI_LANES_READLANE64
#define MI_LANES_READLANE64 1  // 0: whole-wave groups use ds_bpermute too (developer A/B)
#endif
// Entry I (a compile-time index) of the caller's group's chunk register, without a trip through the LDS crossbar
// where the hardware offers one: a DPP row broadcast (row_newbcast, a modifier of a VALU move, 16-lane rows) for
// groups of 8 and 16 lanes (and of 32 with SPLIT32) — two 8-lane groups share a row and take their halves through
// the bank mask — a quad permute for groups of 4 and 2, a scalar readlane for whole waves; ds_bpermute otherwise
// (32-lane groups without SPLIT32).  Two ds_bpermute per non-zero step
// (column and value) share the LDS data path with the gathers of spmm_ldsb.hip: that, not the LDS reads, bounded its
// first version.  Where the gathers come from the L2s the choice is about the chunk size instead:
// a 32-lane group of spmm_group_kernel runs faster on 32-entry chunks through ds_bpermute than on 16-entry chunks
// through DPP (config C2: 0.264 vs 0.289 ms) — fewer trips to memory for col / val per row.
template <int G, int I, bool SPLIT32, typename T>
__device__ __forceinline__ T group_lane(T x) {
  static_assert(sizeof(T) == 4, "one dword per lane");
  const int bits = __builtin_bit_cast(int, x);
  int r;
  if constexpr (G == 16 || (G == 32 && SPLIT32)) {
    r = __builtin_amdgcn_update_dpp(0, bits, 0x150 + I, 0xf, 0xf, true);
  } else if constexpr (G == 8) {
    r = __builtin_amdgcn_update_dpp(0, bits, 0x150 + I, 0xf, 0x3, false);      // lanes 0-7 of every row: their entry I
    r = __builtin_amdgcn_update_dpp(r, bits, 0x150 + 8 + I, 0xf, 0xc, false);  // lanes 8-15: theirs
  } else if constexpr (G == 4) {
    r = __builtin_amdgcn_update_dpp(0, bits, I * 0x55, 0xf, 0xf, true);  // quad_perm:[I,I,I,I]
  } else if constexpr (G == 2) {
    r = __builtin_amdgcn_update_dpp(0, bits, I | (I << 2) | ((2 + I) << 4) | ((2 + I) << 6), 0xf, 0xf, true);  // [I,I,2+I,2+I]
  } else if constexpr (G == 1) {
    r = bits;
  } else if constexpr (G == 64 && MI_LANES_READLANE64) {
    r = __builtin_amdgcn_readlane(bits, I);
  } else {
    r = __shfl(bits, I, G);
  }
  return __builtin_bit_cast(T, r);
}

}  // namespace mi

#endif  // MI_LANES_H_
