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
// component (`dc01`/`dc23`, four wrapping 16-bit sums), which lets the write pass emit absolute DC
// values and removes the separate DC prefix-sum pass over the coefficient buffer
// (src/decode_dc.cu:88-169, which also accumulates in int16).
//
// The loop is written for a wave of 64 lanes in lock-step: what one lane needs every lane pays for,
// so the common path is branch-light, the next bitstream word is fetched one refill ahead, and the
// long-code path has no chain of dependent table reads (see jg_defs.h for the table pack).
#ifndef JG_HUFF_CORE_H_
#define JG_HUFF_CORE_H_

#include "jg_defs.h"

namespace jg {

struct LaneState {
    int p; // bit position relative to the segment
    int n; // coefficient slots committed by the subsequence being decoded
    int c; // data unit index inside the MCU
    int z; // zig-zag index
    uint32_t dc01; // wrapping 16-bit sums of committed DC differences: component 0 | component 1 << 16
    uint32_t dc23; // component 2 | component 3 << 16
};

JG_HD inline uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    // two independent 16-bit lanes, no carry between them (v_pk_add_u16 on gfx950)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 r = __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b);
    return __builtin_bit_cast(uint32_t, r);
#else
    return ((a + b) & 0xFFFFu) | (((a >> 16) + (b >> 16)) << 16);
#endif
}

/// MSB-first bit window over 32-bit words of the segment's destuffed data: two consecutive words `hi`, `lo` and
/// the number of bits `sh` by which the pair has to be shifted right for the window's 32 bits to land in the low
/// word (0..31; the window starts 32 - sh bits into `hi`, at sh == 0 at the first bit of `lo`). Looking at the
/// window is one funnel shift (v_alignbit_b32), consuming bits one subtraction, and a refill -- which a wave of 64
/// lanes executes in nearly every iteration of the symbol loop, because some lane always needs one -- two moves and
/// two adds: `hi = lo, lo = next word, sh += 32`, and the step to the following word. (A 64-bit window that ORs
/// the shifted word in costs a 64-bit shift, two ORs and the bookkeeping of the bit count on top.)
///
/// A `Fetch` walks the words in order: `start(w)` gives the position of word `w` of the segment, `load(pos)`
/// issues the load, `advance(pos)` steps to the next word, and `cook(raw, pos)` turns the loaded value into the
/// word as the window wants it (the host twin reads zero past the segment's padded end as the reference does,
/// decode_huffman_reader.hpp:110-152; the device fetch does not need to, see GlobalFetch). `next_raw` always holds
/// the word at `pos`, fetched one refill before it is cooked and moved in: nothing touches the loaded value in
/// between, so the load's latency is covered by the symbols decoded meanwhile.
template <class Fetch>
struct BitWindow {
    uint32_t hi, lo;
    int sh;
    typename Fetch::Pos pos;
    uint32_t next_raw;

    /// Window at bit `p` (>= 0) of the segment. The pair starts at the word that holds bit p - 1, so that sh stays
    /// in 0..31; at p == 0 that is the word in front of the segment, which is loaded but never looked at.
    JG_HD inline void seek(int p, const Fetch& fetch)
    {
        const int q = p - 1;
        sh          = 31 - (q & 31);
        pos         = fetch.start(q >> 5); // arithmetic shift: -1 for q == -1
        hi          = fetch.cook(fetch.load(pos), pos);
        fetch.advance(pos);
        lo = fetch.cook(fetch.load(pos), pos);
        fetch.advance(pos);
        next_raw = fetch.load(pos);
#if defined(__HIP_DEVICE_COMPILE__)
        // `hi` and `lo` must have ARRIVED before the symbol loop starts. Left in flight, the compiler's wait for
        // them lands on their first use inside the loop -- an s_waitcnt vmcnt(1) in every iteration, which also
        // waits for whatever the iteration before stored (gfx9 has one counter for loads and stores): the write
        // pass then sat out the full latency of its ring flushes (measured: a third of its time). Two moves the
        // compiler cannot look through make it wait here, once.
        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(hi), "+v"(lo));
#endif
    }
    JG_HD inline uint32_t peek(const Fetch& fetch)
    {
        if (sh < 0) { // at most 32 bits are consumed between two looks: one step is enough
#if defined(__HIP_DEVICE_COMPILE__)
            // Two real moves at THIS point, in place: left to itself the register coalescer lets `lo` and `next_raw`
            // share a register, loads the new word into a temporary and copies it over at the loop's back edge --
            // behind an s_waitcnt for the load issued a few instructions earlier, which exposes the whole memory
            // latency in every iteration (measured: write pass +21 %). With the moves pinned here the load below
            // targets next_raw's own register and is waited for one refill later. Both operands are read-write: as
            // fresh values the compiler merges them with the untouched `hi` / `lo` of the lanes that did not refill
            // through two extra register copies per symbol.
            asm("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2" : "+v"(hi), "+v"(lo) : "v"(fetch.cook(next_raw, pos)));
#else
            hi = lo;
            lo = fetch.cook(next_raw, pos);
#endif
            sh += 32;
            fetch.advance(pos);
            next_raw = fetch.load(pos);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_alignbit(hi, lo, static_cast<uint32_t>(sh));
#else
        return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> sh);
#endif
    }
    JG_HD inline void skip(int len) { sh -= len; }
};

/// Pointers into the scan's table pack. On the device the pack lives in LDS and the offsets a kernel hands to the
/// symbol loop (cursor ring position, and the offsets inside the cursor entries, which the kernel patches after
/// loading the pack) are absolute LDS addresses: an address is then a register as it stands, where `base + offset`
/// costs a VALU add per table access -- the base of dynamic LDS is a link-time symbol the compiler does not fold.
/// Only the kernels' translation unit asks for this (JG_TABS_IN_LDS); the parser builds tables with plain pointers.
#if defined(JG_TABS_IN_LDS)
#define JG_TAB_AS __attribute__((address_space(3)))
#define JG_TAB_AT(tabs, off) (reinterpret_cast<::jg::TabPtr>(static_cast<uintptr_t>(static_cast<uint32_t>(off))))
#else
#define JG_TAB_AS
#define JG_TAB_AT(tabs, off) ((tabs) + (off))
#endif
typedef JG_TAB_AS const uint8_t* TabPtr;

JG_HD inline uint32_t ld_u16(TabPtr p) { return *reinterpret_cast<JG_TAB_AS const uint16_t*>(p); }
JG_HD inline uint32_t ld_u32(TabPtr p) { return *reinterpret_cast<JG_TAB_AS const uint32_t*>(p); }

/// Code longer than the first-level LUT: find its length by counting thresholds (all eight
/// thresholds come from one 16-byte read), then one huffval read. Reproduces the reference's
/// `get_category` (src/decode_huffman.cu:167-194): the 16-bit candidate always accepts and the
/// huffval index is reduced modulo 256, so an invalid code still consumes 9..16 bits.
JG_HD inline uint32_t huff_long_code(TabPtr aux, uint32_t peek, bool is_dc)
{
    const uint32_t v = peek >> 16;
    int l            = 9;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 lim = *reinterpret_cast<JG_TAB_AS const u32x4*>(aux); // one 16-byte LDS read
#pragma unroll
    for (int j = 0; j < 7; ++j) l += v >= ((lim[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) ? 1 : 0;
#else
    for (int j = 0; j < 7; ++j) l += v >= ld_u16(aux + 2 * j) ? 1 : 0;
#endif
    const uint32_t code = v >> (16 - l);
    const uint32_t off  = ld_u16(aux + 16 + 2 * (l - 9));
    const uint32_t sym  = aux[32 + ((off + code) & 0xFFu)];
    return huff_entry(l, sym, is_dc);
}

/// First-level entry without a length: second-level table if the host built one for this prefix,
/// else the long-code path.
template <int kEntryBytes = 2>
JG_HD inline uint32_t huff_second_level(TabPtr tab, uint32_t e, uint32_t peek, bool is_dc)
{
    const int lb       = is_dc ? kLutBitsDc : kLutBitsAc;
    const TabPtr aux   = tab + (is_dc ? (kEntryBytes << kLutBitsDc) : (kEntryBytes << kLutBitsAc));
    if (e != 0) {
        const uint32_t i2 = (peek >> (32 - kSubBits - lb)) & ((1u << kSubBits) - 1u);
        e                 = ld_u16(aux + kHuffAuxSize - kSubTableSize + (e >> 5) * kSubTableSize + 2 * i2);
        if (e != 0) return e;
    }
    return huff_long_code(aux, peek, is_dc);
}

JG_HD inline int extend_magnitude(uint32_t bits, int s)
{
    // T.81 F.2.2.1 EXTEND; reference get_value (decode_huffman.cu:196-200) without the signed shift: a magnitude
    // whose first bit is clear stands for bits - (2^s - 1). Written around the mask 2^s - 1 (one bit-field-mask
    // instruction on the device).
    const uint32_t mask = (1u << s) - 1u;
    return bits > (mask >> 1) ? static_cast<int>(bits) : static_cast<int>(bits - mask);
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

/// Sink of the flow passes: nothing is stored; n and the DC sums of the subsequence are accumulated.
struct NoSink {
    static constexpr bool kSums = true;
};

/// Sink of the speculative pass: only the exit state (p, c, z) is wanted -- every subsequence is decoded
/// again from a real predecessor state (or its segment's start) by a flow, which supplies n and the sums.
struct SpecSink : NoSink {
    static constexpr bool kSums = false;
};

/// Entries of the symbol stream the write pass emits (jg_defs.h): the AC entry of a coefficient at zig-zag index
/// `zpos` (value in the high 10 bits, index in the low 6: the store keeps 16 bits, so no mask is needed), the escape
/// entry that follows it if the value does not fit (index 0, the value's bits 10.. in the high bits), and the two
/// ways back.
JG_HD inline uint32_t sym_entry_ac(int zpos, int value) { return (static_cast<uint32_t>(value) << 6) | static_cast<uint32_t>(zpos); }
JG_HD inline uint32_t sym_entry_escape(int value) { return (static_cast<uint32_t>(value) >> 10) << 6; }
/// A coefficient of category 10 or more (|value| >= 512) takes an escape entry; the write pass asks the category.
JG_HD inline bool sym_needs_escape(int value) { return static_cast<uint32_t>(value + 511) > 1022u; }
constexpr int kEscapeFromCategory = 10;
JG_HD inline uint32_t sym_entry_index(uint32_t e) { return e & 63u; }
JG_HD inline int sym_entry_value(uint32_t e) { return static_cast<int16_t>(e) >> 6; }
JG_HD inline int sym_entry_value(uint32_t e, uint32_t escape) { return static_cast<int16_t>(((escape >> 6) << 10) | ((e & 0xFFFFu) >> 6)); }
/// Bit of a data-unit record's count that says "this unit holds an escape" (a count is at most 127).
constexpr uint32_t kUnitHasEscape = 0x80u;

#if defined(__HIP_DEVICE_COMPILE__)
#define JG_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define JG_WAVE_ANY(x) (x) // a host emulation runs one lane at a time
#endif

/// STATE-ONLY decode (speculative pass, flows) from `st` up to bit `end_bit` of the segment, committing the symbols
/// that end at or before it. `st.n`, `st.dc01`, `st.dc23` accumulate (not with SpecSink). `tabs` is the scan's SYNC
/// table pack (LDS on the device). The write pass has its own loop (decode_units, below).
///
/// ONE flat loop for every lane: the data-unit boundary is handled with selects, not with a branch -- a branch
/// there makes the compiler nest the loop, and a nested loop makes the 64 lanes of a wave wait for the longest
/// data unit among them at every boundary. Coefficient slots are counted as 64 per closed unit plus the zig-zag
/// index difference, which equals the reference's per-symbol count (decode_huffman.cu:302-394) on every valid stream.
template <class Fetch, class Sink>
JG_HD inline void decode_subsequence(
    LaneState& st,
    BitWindow<Fetch>& bw,
    const Fetch& fetch,
    int end_bit,
    const uint8_t* tabs,
    const ScanParams& sp,
    Sink&)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define JG_LOAD_CURSOR(off) (*reinterpret_cast<JG_TAB_AS const u32x4*>(JG_TAB_AT(tabs, off)))
    u32x4 cur = JG_LOAD_CURSOR(sp.cursor_off + 16u * static_cast<uint32_t>(st.c));
#define JG_CUR_TABS cur[0]
#define JG_CUR_META cur[1]
#define JG_CUR_SELF cur[2]
#define JG_CUR_NEXT cur[3]
#else
#define JG_LOAD_CURSOR(off) (*reinterpret_cast<const CursorEntry*>(tabs + (off)))
    CursorEntry cur = JG_LOAD_CURSOR(sp.cursor_off + 16u * static_cast<uint32_t>(st.c));
#define JG_CUR_TABS cur.tabs
#define JG_CUR_META cur.meta
#define JG_CUR_SELF cur.self
#define JG_CUR_NEXT cur.next
#endif
    int p       = st.p;
    int zm      = st.z - 1; // zig-zag index MINUS ONE: the index of the coefficient a symbol ends at is zm + its advance
    int units   = 0;
    uint32_t dc01 = st.dc01, dc23 = st.dc23;
    bool is_dc = st.z == 0;
    uint32_t peek, e;
    int total;
    // commit the symbol: advance the window and the zig-zag position, feed the sink / the sums, move the cursor
#define JG_COMMIT()                                                                                       \
    do {                                                                                                  \
        bw.skip(total);                                                                                   \
        p += total;                                                                                       \
        const int adv     = e >> 9;                                                                       \
        const int zp      = zm + adv; /* index of the symbol's coefficient */                             \
        const bool du_end = zp >= 63;                                                                     \
        if (Sink::kSums && is_dc) {                                                                       \
            const int s      = (e >> 5) & 15;                                                             \
            const int v      = extend_magnitude(bits_field(peek, total, s), s);                           \
            const uint64_t d = static_cast<uint64_t>(static_cast<uint32_t>(v) & 0xFFFFu) << (JG_CUR_META & 63); \
            dc01             = pk_add_u16(dc01, static_cast<uint32_t>(d));                                \
            dc23             = pk_add_u16(dc23, static_cast<uint32_t>(d >> 32));                          \
        }                                                                                                 \
        zm = du_end ? -1 : zp;                                                                            \
        if (Sink::kSums) units += du_end ? 1 : 0;                                                         \
        cur = JG_LOAD_CURSOR(du_end ? JG_CUR_NEXT : JG_CUR_SELF);                                         \
        is_dc = du_end;                /* a unit just ended <=> the next symbol is a DC symbol */         \
    } while (0)
    {
        // State-only passes walk the SYNC pack (jg_defs.h): 32-bit first-level entries whose high half stands for
        // as many AC symbols as lie inside the index bits. While at least 31 bits are left in front of `end_bit`
        // whatever an entry stands for fits (a symbol takes at most 16 + 15 bits, a multi-symbol entry at most the 11
        // index bits), so the main loop does not ask; it takes the high half unless the data unit would end in
        // front of the last of its symbols -- index + advance of the earlier ones reaching 64: the following
        // symbol is then a DC symbol of the next unit, not what the AC table made of those bits -- and the low
        // half, the first symbol alone, otherwise. The symbols committed are exactly those of the one-symbol-per-
        // step loop, which finishes the subsequence: it looks the next symbol up at the END of the body and tests
        // it in the loop condition (one compare and one conditional back edge for "does the symbol still fit").
        // In both loops the DC / AC choice of a symbol is the unit-end flag of the one before it.
        while (end_bit - p >= 31) {
            peek               = bw.peek(fetch);
            const TabPtr tab   = JG_TAB_AT(tabs, is_dc ? (JG_CUR_TABS & 0xFFFFu) : (JG_CUR_TABS >> 16));
            const uint32_t idx = peek >> (is_dc ? 32 - kLutBitsDc : 32 - kLutBitsAc);
            const uint32_t e32 = ld_u32(tab + kSyncEntryBytes * idx);
            // the choice between the halves with selects, the second level -- a code longer than the index bits, rare --
            // behind ONE question to the whole wave: as `if long code ... else choose` every step paid two exec-mask
            // hand-offs between the vector and the scalar unit whether any lane had a long code or not
            const uint32_t one = e32 & 0xFFFFu, m = e32 >> 16;
            e                  = zm + static_cast<int>((m >> 5) & 15u) < 63 ? m : one;
            if (JG_WAVE_ANY((one & 31u) == 0)) {
                if ((one & 31u) == 0) e = huff_second_level<kSyncEntryBytes>(tab, one, peek, is_dc);
            }
            total = e & 31;
            JG_COMMIT();
        }
#define JG_LOOKUP_SYNC()                                                                                  \
    do {                                                                                                  \
        peek               = bw.peek(fetch);                                                              \
        const TabPtr tab   = JG_TAB_AT(tabs, is_dc ? (JG_CUR_TABS & 0xFFFFu) : (JG_CUR_TABS >> 16));      \
        const uint32_t idx = peek >> (is_dc ? 32 - kLutBitsDc : 32 - kLutBitsAc);                         \
        e                  = ld_u16(tab + kSyncEntryBytes * idx);                                         \
        if ((e & 31u) == 0) e = huff_second_level<kSyncEntryBytes>(tab, e, peek, is_dc);                  \
        total = e & 31;                                                                                   \
    } while (0)
        JG_LOOKUP_SYNC();
        while (p + total <= end_bit) {
            JG_COMMIT();
            JG_LOOKUP_SYNC();
        }
#undef JG_LOOKUP_SYNC
    }
#undef JG_COMMIT
    if (Sink::kSums) st.n += 64 * units + (zm + 1) - st.z;
    st.p    = p;
    st.z    = zm + 1;
    st.c    = (JG_CUR_META >> 8) & 0xFF;
    st.dc01 = dc01;
    st.dc23 = dc23;
#undef JG_LOAD_CURSOR
#undef JG_CUR_TABS
#undef JG_CUR_META
#undef JG_CUR_SELF
#undef JG_CUR_NEXT
}

/// Iterations of the write pass's loop between two DC SLOTS (below).
#ifndef JG_WRITE_DC_PERIOD
#define JG_WRITE_DC_PERIOD 4
#endif
constexpr int kWriteDcPeriod = JG_WRITE_DC_PERIOD; // a power of two
/// ... and between two RARE SLOTS.
#ifndef JG_WRITE_RARE_PERIOD
#define JG_WRITE_RARE_PERIOD 8
#endif
constexpr int kWriteRarePeriod = JG_WRITE_RARE_PERIOD; // a power of two

JG_HD inline uint32_t bit_mask(int s)
{
    // 2^s - 1, s in 0..31 (one v_bfm_b32 on the device; the compiler makes a shift and a NOT of the C form)
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m;
    asm("v_bfm_b32 %0, %1, 0" : "=v"(m) : "v"(s));
    return m;
#else
    return (1u << s) - 1u;
#endif
}

/// First-level entry (16 bits) of the table at `tab` for the window `peek`: entry (peek >> (32 - kLutBits)). On the
/// device the address is a shift and a shift-add; left alone the compiler folds the two shifts into shift, AND, add.
template <int kLutBits>
JG_HD inline uint32_t lut16_entry(TabPtr tab, uint32_t peek)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(JG_TABS_IN_LDS)
    uint32_t a;
    asm("v_lshrrev_b32 %0, %1, %2\n\tv_lshl_add_u32 %0, %0, 1, %3"
        : "=&v"(a)
        : "n"(32 - kLutBits), "v"(peek), "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab))));
    return ld_u16(reinterpret_cast<TabPtr>(static_cast<uintptr_t>(a)));
#else
    return ld_u16(tab + 2 * (peek >> (32 - kLutBits)));
#endif
}

/// First-level AC entry of the WRITE pack as a SIGNED 16-bit value: negative where the entry carries kEntrySlow (jg_defs.h).
JG_HD inline int lut16_entry_signed(TabPtr tab, uint32_t peek)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(JG_TABS_IN_LDS)
    uint32_t a;
    asm("v_lshrrev_b32 %0, %1, %2\n\tv_lshl_add_u32 %0, %0, 1, %3"
        : "=&v"(a)
        : "n"(32 - kLutBitsAc), "v"(peek), "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab))));
    return *reinterpret_cast<JG_TAB_AS const int16_t*>(static_cast<uintptr_t>(a)); // ds_read_i16
#else
    return *reinterpret_cast<JG_TAB_AS const int16_t*>(tab + 2 * (peek >> (32 - kLutBitsAc)));
#endif
}

/// `e`, or 0 where `x` is negative: as a shift and a bit-field insert on the device (the compiler turns the C form back into
/// compare + select, which costs the wave a wait state between the two on gfx950 and the vcc register).
JG_HD inline uint32_t zero_if_negative(uint32_t e, int x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1\n\tv_bfi_b32 %0, %0, 0, %2" : "=&v"(r) : "v"(x), "v"(e));
    return r;
#else
    return x < 0 ? 0u : e;
#endif
}

/// EXTEND of the `s` bits that end `total` bits into the window (s == 0 gives 0). On the device: the field as a SIGNED
/// bit-field t (negative where the magnitude's first bit is set: t = bits - 2^s), then t - ((2^s - 1) ^ (t >> 31)): bits
/// for a set first bit, bits - (2^s - 1) for a clear one -- no compare, no select (a select behind a compare costs the
/// wave a wait state on gfx950).
JG_HD inline int extend_field(uint32_t peek, int total, int s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int t = __builtin_amdgcn_sbfe(static_cast<int>(peek), static_cast<uint32_t>(32 - total), static_cast<uint32_t>(s));
    return t - static_cast<int>(bit_mask(s) ^ static_cast<uint32_t>(t >> 31));
#else
    const uint32_t bits = bits_field(peek, total, s);
    const uint32_t mask = bit_mask(s);
    return bits > (mask >> 1) ? static_cast<int>(bits) : static_cast<int>(bits - mask);
#endif
}

/// T.81 F.2.2.1 EXTEND as extend_magnitude above, around bit_mask.
JG_HD inline int extend_bits(uint32_t bits, int s)
{
    const uint32_t mask = bit_mask(s);
    return bits > (mask >> 1) ? static_cast<int>(bits) : static_cast<int>(bits - mask);
}


/// The symbol loop of the WRITE PASS: decode from `st` and hand every coefficient to `sink`, for the data units the
/// lane OWNS: those whose DC symbol its subsequence commits. It runs past the end of the subsequence to finish the last
/// of them and stops in front of the first unit of the next lane -- which the caller knows as a unit INDEX (the
/// coefficient counts of the synchronisation passes place every lane's first unit: `sink.full()`), so no bit position
/// is tracked here at all. Same symbols, same rules as decode_subsequence (reference decode_huffman.cu:302-394,
/// 627-682), arranged for what bounds such a loop on gfx950: not the number of vector instructions alone but the
/// hand-offs between the vector and the scalar unit -- every `if` on a per-lane condition is a vector compare, a
/// scalar exec-mask update and usually a branch, ~35 cycles before the next vector instruction of the wave can go
/// (tools/probe/chain_probe.hip; DESIGN.md section 3 has the measurements behind this arrangement).
///
///   * ONE uniform loop: lanes never leave it alone. A lane that is done, a lane that waits, a lane whose look-up
///     needs the slow path all run the same instructions on a NULL table entry (length, category and advance 0),
///     which changes nothing; the wave leaves when every lane has stopped.
///   * The AC step of every iteration has no per-lane branch at all: the window refill is three selects, the reload
///     of the prefetched word an exec-masked load, the entry is nulled by one select on the sign of an OR of the
///     reasons a lane may not step (outside a unit, window not yet refilled, a symbol for the rare slot).
///   * Everything rare -- second-level look-up / long code, a coefficient of category >= 10 with its escape entry, the
///     window leaving its row -- WAITS for a rare slot, every kWriteRarePeriod-th iteration, and only there the wave
///     asks whether any lane has such a thing. A lane meets one less than three times per 256-byte subsequence, so the
///     waiting is cheap (3.5 iterations each); asked in every iteration the wave found some lane's in 27 % of them and
///     ran the block's ~40 instructions (tools/probe/write_lane_iters.py).
///   * Every kWriteDcPeriod-th iteration has a DC SLOT: the lanes that stand at the start of a data unit close the
///     record of the unit they finished, test the stop rule, load the next unit's cursor entry (tables, component),
///     decode the DC symbol and add it to the component's predictor. A lane that reaches a unit's end between two
///     slots WAITS for the next one (1.5 iterations per unit of ~15 symbols): in a wave of 64 lanes some lane is at a
///     unit boundary in nearly every iteration, so a loop that handles boundaries where they fall pays that work
///     -- about as many instructions as the AC work -- every iteration.
///
/// `Window` is the lane's view of the bitstream (device: RowWindow over the tiled rows of the destuffed buffer,
/// jg_kernels.hip; host twin: tests/emu): seek(p), top() -- the refill, once per iteration --, look() -- the 32 bits
/// at the position, valid while left() >= 0 --, skip(n), left() -- negative when the window wants a refill --,
/// crossed() -- negative where the position has just left its row --, cross() and done() behind the loop. The results
/// do not depend on the slot periods (a host emulation runs one lane at a time). `max_iters` bounds the loop whatever
/// the stream holds; a valid one needs at most kWriteDcPeriod iterations per two bits (a data unit of a one-bit DC code
/// and a one-bit end of block waits for its DC slot: jg_kernels.hip derives the bound from the periods).
template <class Window, class Sink>
JG_HD inline void decode_units(
    const LaneState& st,
    Window& w,
    const uint8_t* tabs,
    const ScanParams& sp,
    Sink& sink,
    int max_iters,
    int* iters_out = nullptr) // probe builds: [0] iterations the lane's wave stayed in the loop, [1] symbols the lane decoded,
                              // [2] times the wave took the rare block, [3] times this lane was a reason for it
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define JG_LOAD_CURSOR(off) (*reinterpret_cast<JG_TAB_AS const u32x4*>(JG_TAB_AT(tabs, off)))
#define JG_CUR_TABS cur[0]
#define JG_CUR_META cur[1]
#define JG_CUR_SELF cur[2]
#define JG_CUR_NEXT cur[3]
    typedef u32x4 Cursor;
#else
#define JG_LOAD_CURSOR(off) (*reinterpret_cast<const CursorEntry*>(tabs + (off)))
#define JG_CUR_TABS cur.tabs
#define JG_CUR_META cur.meta
#define JG_CUR_SELF cur.self
#define JG_CUR_NEXT cur.next
    typedef CursorEntry Cursor;
#endif
    // zig-zag index minus one of the lane's position inside its unit, 0..62; 63..127: the unit is complete and the
    // next symbol is a DC symbol (kAtUnitStart for a lane that starts there); kStopped: the lane is done.
    constexpr int kAtUnitStart = 64, kStopped = 128;
    int zm;
    uint32_t actab, unit_entry; // AC table of the unit the lane is in; cursor entry of the next unit to start
    {
        const Cursor cur = JG_LOAD_CURSOR(sp.cursor_off + 16u * static_cast<uint32_t>(st.c));
        zm               = st.z ? st.z - 1 : kAtUnitStart;
        actab            = JG_CUR_TABS >> 16;
        unit_entry       = st.z ? JG_CUR_NEXT : JG_CUR_SELF;
    }
    uint32_t dc01 = st.dc01, dc23 = st.dc23; // predictors: running sums of the DC differences, per component
    w.seek(st.p);
    int flush_in = Sink::kFlushPeriod; // iterations to the sink's next flush point
    int flush_no = 0;                  // how many there have been
    // The loop is UNROLLED over the DC period (round 5): one body = the DC slot and kWriteDcPeriod AC steps, then -- every
    // kRareBodies-th body -- the rare slot, and -- every kFlushBodies-th -- the flush point. As a loop over single iterations
    // every one of them paid three scalar tests with their branches (`it & 3`, `it & 7`, the flush counter: a third of the
    // loop's scalar instructions, and scalar instructions are not free riders in a loop bound by instruction issue:
    // DESIGN.md section 3); now a body pays two. The slots fall on the same iterations as before.
    static_assert(kWriteRarePeriod % kWriteDcPeriod == 0 && Sink::kFlushPeriod % kWriteDcPeriod == 0, "slot periods are whole bodies");
    constexpr int kRareBodies = kWriteRarePeriod / kWriteDcPeriod, kFlushBodies = Sink::kFlushPeriod / kWriteDcPeriod;
    static_assert((kRareBodies & (kRareBodies - 1)) == 0, "a power of two");
    flush_in = kFlushBodies; // bodies to the sink's next flush point
    // what the rare slot needs of the AC step in front of it
    uint32_t peek = 0;
    int e0 = 0, ready = 0; // the step's entry, signed (kEntrySlow is the sign); >= 0: the lane stands inside a unit with bits in its window
    uint32_t tab_off = 0;
    // AC step, every lane
    const auto ac_step = [&]() __attribute__((always_inline)) {
        peek             = w.look();
        tab_off          = actab;
        const TabPtr tab = JG_TAB_AT(tabs, actab);
        e0               = lut16_entry_signed(tab, peek);
        // No step for a lane outside a unit (zm >= 63) or with a window that ran out (the DC symbol just emptied it, or
        // its refill has to wait an iteration): `ready`. None here for an entry marked slow -- no length, or a category
        // with an escape entry -- or with a window about to leave its row: the lane waits for the rare slot. The sign
        // of one OR says so, and the entry is nulled by that sign (no compare, no select).
        ready            = (62 - zm) | w.left();
        const uint32_t e = zero_if_negative(static_cast<uint32_t>(e0), ready | e0 | w.crossed());
        const int total  = e & 31;
        if (iters_out && e != 0) ++iters_out[1];
        w.skip(total);
        const int s = (e >> 5) & 15;
        zm += static_cast<int>(e >> 9); // index of the symbol's coefficient; 63 or more: the unit is complete
        sink.ac(s, zm, extend_field(peek, total, s));
    };
    int body = 0;
    for (int it = 0; it < max_iters; it += kWriteDcPeriod, ++body) {
        if (iters_out) iters_out[0] = it + kWriteDcPeriod;
        w.top(); // refill where the window ran out (at most 32 bits are consumed between two looks)
        { // DC slot: the same iterations for every lane of a wave
            // (a lane whose window is about to leave its row first waits for the rare slot: the DC symbol may empty the
            // window, and the refill behind it would step out of the row)
            if (static_cast<uint32_t>(zm - 63) < static_cast<uint32_t>(kStopped - 63) && (w.left() | w.crossed()) >= 0) {
                sink.unit_boundary(); // the unit the lane finished since the last slot, if any, is complete
                if (sink.full()) {
                    zm = kStopped; // the next unit is the next lane's, or lies past the segment
                } else {
                    const Cursor cur    = JG_LOAD_CURSOR(unit_entry);
                    const uint32_t dpeek = w.look();
                    const TabPtr tab    = JG_TAB_AT(tabs, JG_CUR_TABS & 0xFFFFu);
                    uint32_t e          = lut16_entry<kLutBitsDc>(tab, dpeek);
                    if ((e & 31u) == 0) e = huff_second_level(tab, e, dpeek, true);
                    const int total = e & 31;
                    w.skip(total); // the window may run out: the lane then sits the AC step of this iteration out
                    const int s      = (e >> 5) & 15;
                    const int v      = extend_bits(bits_field(dpeek, total, s), s);
                    const int csh    = JG_CUR_META & 63;
                    const uint64_t d = static_cast<uint64_t>(static_cast<uint32_t>(v) & 0xFFFFu) << csh;
                    dc01             = pk_add_u16(dc01, static_cast<uint32_t>(d));
                    dc23             = pk_add_u16(dc23, static_cast<uint32_t>(d >> 32));
                    // the component's running sum is the absolute DC value, 16-bit wrap like the reference's int16
                    // prefix sum (decode_dc.cu:129-155)
                    sink.dc(static_cast<int>(((static_cast<uint64_t>(dc23) << 32) | dc01) >> csh));
                    actab      = JG_CUR_TABS >> 16;
                    unit_entry = JG_CUR_NEXT;
                    zm         = 0;
                    if (iters_out) ++iters_out[1];
                }
            }
            if (!JG_WAVE_ANY(zm != kStopped)) break;
        }
        ac_step();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 1; k < kWriteDcPeriod; ++k) {
            w.top();
            ac_step();
        }
        if ((body & (kRareBodies - 1)) == kRareBodies - 1) { // rare slot: the same iterations for every lane of a wave
            // (zm and the window of a lane that waited are what they were: the null entry changed nothing)
            const bool symbol = ready >= 0 && e0 < 0;
            if (JG_WAVE_ANY(symbol || w.crossed() < 0)) {
                if (iters_out) {
                    ++iters_out[2];
                    if (symbol || w.crossed() < 0) ++iters_out[3];
                }
                if (w.crossed() < 0) w.cross(); // the symbol in the window, if it was only this, goes with the next iteration
                if (symbol) {
                    const TabPtr tab  = JG_TAB_AT(tabs, tab_off);
                    const uint32_t e1 = static_cast<uint32_t>(e0) & (kEntrySlow - 1u); // the entry without its mark
                    const uint32_t e2 = (e1 & 31u) == 0 ? huff_second_level(tab, e1, peek, false) : e1;
                    const int total2  = e2 & 31;
                    if (iters_out) ++iters_out[1];
                    w.skip(total2);
                    const int s2 = (e2 >> 5) & 15;
                    zm += static_cast<int>(e2 >> 9);
                    const int v2 = extend_bits(bits_field(peek, total2, s2), s2);
                    sink.ac(s2, zm, v2);
                    if (s2 >= kEscapeFromCategory) sink.escape(v2);
                }
            }
        }
        if (--flush_in == 0) { // the same iteration for every lane of the wave
            flush_in = kFlushBodies;
            sink.flush_point(flush_no++);
        }
    }
    w.done();
#undef JG_LOAD_CURSOR
#undef JG_CUR_TABS
#undef JG_CUR_META
#undef JG_CUR_SELF
#undef JG_CUR_NEXT
}

} // namespace jg

#endif // JG_HUFF_CORE_H_
