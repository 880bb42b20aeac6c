// jg_huff_core.h -- the per-lane Huffman symbol loop, written once for every kernel that walks the
// bitstream (speculative pass, intra-/inter-sequence flows, write pass). Compiles for gfx950 and,
// with JG_HD empty, for the host so tests can emulate the subsequence-parallel algorithm on a CPU.
//
// Semantics follow the reference's `decode_subsequence` (src/decode_huffman.cu:302-394) and
// `decode_next_symbol*` (:202-286), SURVEY.md Appendix E:
//   * lane state between symbols is (p, c, z): bit position in the segment, data-unit index in the
//     MCU, zig-zag index; `n` counts coefficient slots committed;
//   * a subsequence commits the symbols that END at or before its last bit, a straddling symbol
//     belongs to the next subsequence;
//   * AC (0,0) = end of block, (15,0) = 16 zeros, any other s == 0 symbol is treated as end of block;
//   * z >= 64 after a symbol closes the data unit whatever the overshoot (only reachable while
//     decoding from a wrong speculative state or on corrupt input).
// New relative to the reference: the DC differences committed by a subsequence are summed per scan
// component (`dc[]`), which lets the write pass emit absolute DC values and removes the separate
// DC prefix-sum pass over the coefficient buffer (src/decode_dc.cu:88-169).
#ifndef JG_HUFF_CORE_H_
#define JG_HUFF_CORE_H_

#include "jg_defs.h"

namespace jg {

struct LaneState {
    int p; // bit position relative to the segment
    int n; // coefficient slots committed by the subsequence being decoded
    int c; // data unit index inside the MCU
    int z; // zig-zag index
    int dc[kMaxComp]; // sum of committed DC differences per scan component
};

/// MSB-first 64-bit window over big-endian 32-bit words. `Fetch(w)` returns word `w` of the
/// segment's destuffed data (zero past the padded end, reference decode_huffman_reader.hpp:110-152).
template <class Fetch>
struct BitWindow {
    uint64_t win;
    int avail;
    int next_word;

    JG_HD inline void seek(int p, const Fetch& fetch)
    {
        const int w   = p >> 5;
        const int off = p & 31;
        win           = ((static_cast<uint64_t>(fetch(w)) << 32) | fetch(w + 1)) << off;
        avail         = 64 - off;
        next_word     = w + 2;
    }
    JG_HD inline uint32_t peek(const Fetch& fetch)
    {
        if (avail < 32) {
            win |= static_cast<uint64_t>(fetch(next_word)) << (32 - avail);
            avail += 32;
            ++next_word;
        }
        return static_cast<uint32_t>(win >> 32);
    }
    JG_HD inline void skip(int len)
    {
        win <<= len;
        avail -= len;
    }
};

/// Decode one Huffman code from the 32 MSB-aligned bits `peek`. Returns the digested entry
/// (layout: HuffTableDev). Codes longer than 8 bits walk maxcode[] like the reference's
/// `get_category` (src/decode_huffman.cu:167-194): the 16-bit iteration always accepts and the
/// huffval index is reduced modulo 256, so an invalid code still consumes 9..16 bits.
JG_HD inline uint32_t huff_lookup(const HuffTableDev* t, uint32_t peek, bool is_dc)
{
    uint32_t e = t->lut[peek >> 24];
    if ((e & 31u) == 0) {
        int l = 8; // candidate length - 1
        int32_t code;
        for (;; ++l) {
            code = static_cast<int32_t>(peek >> (31 - l));
            if (l == 15 || code <= t->maxcode[l]) break;
        }
        const uint32_t sym = t->huffval[static_cast<uint8_t>(t->valoff[l] + code)];
        e                  = huff_entry(0, l + 1, sym, is_dc);
    }
    return e;
}

JG_HD inline int extend_magnitude(uint32_t bits, int s)
{
    // T.81 F.2.2.1 EXTEND; reference get_value (decode_huffman.cu:196-200) without the signed shift.
    const uint32_t half = (1u << s) >> 1;
    return bits < half ? static_cast<int>(bits - (1u << s) + 1u) : static_cast<int>(bits);
}

JG_HD inline uint32_t bits_field(uint32_t peek, int total_len, int s)
{
    // the s bits that follow the code word; s == 0 gives 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(peek, 32 - total_len, s);
#else
    return s ? (peek << (total_len - s)) >> (32 - s) : 0u;
#endif
}

struct TableSel {
    const HuffTableDev* dc;
    const HuffTableDev* ac;
    int comp;
};

JG_HD inline TableSel select_tables(const HuffTableDev* tables, const ScanParams& sp, int c)
{
    TableSel r;
    r.comp = (sp.du_comp >> (2 * c)) & 3;
    r.dc   = tables + ((sp.dc_slot >> (4 * r.comp)) & 15);
    r.ac   = tables + ((sp.ac_slot >> (4 * r.comp)) & 15);
    return r;
}

/// Sink used by the synchronisation passes: nothing is stored.
struct NoSink {
    static constexpr bool kWrite = false;
    JG_HD inline bool full() const { return false; }
    JG_HD inline void dc(int, int) {}
    JG_HD inline void ac(int, int) {}
    JG_HD inline void advance(int) {}
};

/// Decode subsequence `sub_rel` (index inside its segment) from `st`, committing symbols that end
/// at or before the subsequence's last bit. `st.n` and `st.dc[]` accumulate.
template <class Fetch, class Sink>
JG_HD inline void decode_subsequence(
    LaneState& st,
    BitWindow<Fetch>& bw,
    const Fetch& fetch,
    int end_bit,
    const HuffTableDev* tables,
    const ScanParams& sp,
    Sink& sink)
{
    TableSel ts = select_tables(tables, sp, st.c);
    while (true) {
        if (Sink::kWrite && sink.full()) break;
        const uint32_t peek = bw.peek(fetch);
        const bool is_dc    = st.z == 0;
        const uint32_t e    = huff_lookup(is_dc ? ts.dc : ts.ac, peek, is_dc);
        const int total     = (e >> 5) & 63;
        if (st.p + total > end_bit) break;
        bw.skip(total);
        st.p += total;
        const int s = (e >> 11) & 15;
        int adv;
        if (is_dc) {
            const int diff = extend_magnitude(bits_field(peek, total, s), s);
            // unrolled select instead of dc[comp]: a runtime-indexed register array goes to scratch
            st.dc[0] += ts.comp == 0 ? diff : 0;
            st.dc[1] += ts.comp == 1 ? diff : 0;
            st.dc[2] += ts.comp == 2 ? diff : 0;
            st.dc[3] += ts.comp == 3 ? diff : 0;
            sink.dc(ts.comp, diff);
            adv = 1;
        } else if (e & 0x8000u) {
            adv = 64 - st.z;
            sink.advance(adv);
        } else {
            adv = (e >> 16) & 31;
            if (Sink::kWrite) {
                if (s) sink.ac(adv - 1, extend_magnitude(bits_field(peek, total, s), s));
                else sink.advance(adv);
            }
        }
        st.n += adv;
        st.z += adv;
        if (st.z >= 64) {
            st.z = 0;
            st.c = st.c + 1 >= sp.du_per_mcu ? 0 : st.c + 1;
            ts   = select_tables(tables, sp, st.c);
        }
    }
}

} // namespace jg

#endif // JG_HUFF_CORE_H_
