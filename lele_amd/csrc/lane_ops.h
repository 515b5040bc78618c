// lane_ops.h -- cross-lane moves inside a 32-lane group (a half wave) on the DPP / permlane network instead of ds_bpermute.
// __shfl* compile to ds_bpermute_b32: an LDS-crossbar round trip plus address arithmetic per call, ~20 of them in a dependent
// chain per softmax / LayerNorm row (measured: the softmax phase of the fused attention kernel cost as much as its Q K^T phase).
// The patterns those reductions need are all row-local (DPP, full VALU rate) or a swap of adjacent 16-lane rows
// (v_permlane16_swap_b32, new on gfx950).  Every function documents which lanes of its result are defined; the callers
// (norm_core.h) only consume those.  Checked against __shfl on the device by tools/lane_ops_check.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace lele {

template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {  // lanes without a source inside the row read 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// lane l <- lane l + N of the same 16-lane row (N = 1, 2, 4, 8); lanes whose source falls outside the row read 0
// == __shfl_down(x, N, 32) in the lanes with (l & 15) + N < 16
template <int N>
__device__ __forceinline__ float row_down(float x) { return dpp_move<0x100 + N>(x); }  // row_shl:N

// the value the lane 16 positions away (inside the 32-lane group) holds: even rows <- next row, odd rows <- previous row
// == __shfl_xor(x, 16, 32)
__device__ __forceinline__ float swap16(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // r[0]: odd rows hold the even rows' values; r[1]: even rows hold the odd rows'
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}

// all-reduce over the 32-lane group with a commutative, associative, EXACT operation (min / max: the order of the combination does
// not change the value), result in every lane: quads, half rows, rows on DPP, then the adjacent row
template <class F>
__device__ __forceinline__ float group_allreduce32(float m, F op) {
    auto partner = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    m = op(m, partner(m, std::integral_constant<int, 0xB1>()));   // quad_perm [1,0,3,2]
    m = op(m, partner(m, std::integral_constant<int, 0x4E>()));   // quad_perm [2,3,0,1]
    m = op(m, partner(m, std::integral_constant<int, 0x141>()));  // row_half_mirror
    m = op(m, partner(m, std::integral_constant<int, 0x140>()));  // row_mirror
    return op(m, swap16(m));
}
__device__ __forceinline__ float group_max32(float m) {  // fmaxf semantics
    return group_allreduce32(m, [](float a, float b) { return fmaxf(a, b); });
}

// the value the lane 32 positions away holds (the other half of the wave)  == __shfl_xor(x, 32)
__device__ __forceinline__ float swap32(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0]: the upper half holds the lower half's values; r[1]: the reverse
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
// all-reduce over the whole wave (same contract as group_allreduce32)
template <class F>
__device__ __forceinline__ float wave_allreduce64(float m, F op) {
    m = group_allreduce32(m, op);
    return op(m, swap32(m));
}
// exact integer sum over the wave, in every lane
__device__ __forceinline__ int wave_sum_i32(int v) {
    auto add = [](float a, float b) { return __int_as_float(__float_as_int(a) + __float_as_int(b)); };
    return __float_as_int(wave_allreduce64(__int_as_float(v), add));
}

// the value lane `idx` (0..31, wave-uniform) of the caller's 32-lane group holds, in every lane  == __shfl(x, idx, 32)
__device__ __forceinline__ float group_read(float x, int idx) {
    const int lo = __builtin_amdgcn_readlane(__float_as_int(x), idx), hi = __builtin_amdgcn_readlane(__float_as_int(x), idx + 32);
    return __int_as_float((threadIdx.x & 32) ? hi : lo);
}

}  // namespace lele
