// jg_bytes.h -- the byte rule of the entropy-coded segment on four bytes at a time (device code; compiles for the host
// as well, so that tests can push the reference-built known answers through it: tests/test_huff_kats.py).
//
// Reference src/decode_destuff.cu:37-44: a byte is data iff (prev == FF and b == 00) -- it stands for the FF in
// front of it -- or (prev != FF and b != FF). The kernels that walk the stuffed bytes (destuff_kernel,
// jg_front.hip) own 16 bytes per lane as four little-endian words and classify them with masks that carry
// one bit per byte at bit 7 of that byte ("0x80 domain") instead of sixteen scalar byte tests.
#ifndef JG_BYTES_H_
#define JG_BYTES_H_

#include <cstdint>

#if defined(__HIPCC__)
#define JG_BYTES_FN __device__ __forceinline__
#else
#define JG_BYTES_FN inline
#endif

namespace jg {

constexpr uint32_t kHi80 = 0x80808080u;

/// 0x80 in every byte of `x` that equals FF (exact: the low seven bits + 1 reach bit 7 only for 7F, no carry
/// leaves a byte).
JG_BYTES_FN uint32_t bytes_ff(uint32_t x) { return x & ((x & 0x7F7F7F7Fu) + 0x01010101u) & kHi80; }

/// 0x80 in every byte of `x` that equals 00 (exact).
JG_BYTES_FN uint32_t bytes_zero(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & kHi80; }

/// Bits 7, 15, 23, 31 -> bits 0..3.
JG_BYTES_FN uint32_t collapse80(uint32_t m)
{
    uint32_t x = m >> 7;
    x |= x >> 7;
    x |= x >> 14;
    return x & 0xFu;
}

/// 0x80-domain mask -> FF in the marked bytes.
JG_BYTES_FN uint32_t spread80(uint32_t m) { return m | (m - (m >> 7)); }

/// (hi:lo >> n, low word: v_alignbit_b32 on the device)
JG_BYTES_FN uint32_t funnel_right(uint32_t hi, uint32_t lo, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, n);
#else
    return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> n);
#endif
}
/// Mask of a word whose byte k says what the same mask says about the byte in FRONT of byte k; `before` is the
/// mask of the previous word.
JG_BYTES_FN uint32_t of_previous_byte(uint32_t m, uint32_t before) { return funnel_right(m, before, 24); }
/// ... about the byte BEHIND byte k; `after` is the mask of the next word.
JG_BYTES_FN uint32_t of_next_byte(uint32_t m, uint32_t after) { return funnel_right(after, m, 8); }

} // namespace jg

#endif // JG_BYTES_H_
