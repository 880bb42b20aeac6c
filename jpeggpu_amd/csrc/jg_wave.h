// jg_wave.h -- wave64 helpers for gfx950 (device code).
#ifndef JG_WAVE_H_
#define JG_WAVE_H_

#include <cstdint>

namespace jg {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

/// Inclusive prefix sum over the 64 lanes of a wave with DPP adds: shifts inside the rows of 16 lanes, then the
/// last lane of a row is broadcast into the rows behind it (row_bcast:15 into rows 1 and 3, row_bcast:31 into
/// rows 2 and 3). Lanes without a source add the `old` operand, 0. Six adds instead of six ds_bpermute round trips.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const auto dpp = [](uint32_t x, auto ctrl, auto rows) {
        return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), decltype(ctrl)::value, decltype(rows)::value, 0xF, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xF>{}); // row_shr:1
    v += dpp(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xF>{}); // row_shr:2
    v += dpp(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xF>{}); // row_shr:4
    v += dpp(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xF>{}); // row_shr:8
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{}); // row_bcast:15 -> rows 1, 3
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{}); // row_bcast:31 -> rows 2, 3
    return v;
}

} // namespace jg

#endif // JG_WAVE_H_
