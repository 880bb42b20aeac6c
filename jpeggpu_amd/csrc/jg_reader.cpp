// jg_reader.cpp -- see jg_reader.hpp. Status codes per defect follow src/reader.cpp of the reference.
#include "jg_reader.hpp"
#include "jg_huff_core.h"

#include <algorithm>
#if defined(__SSE2__)
#include <immintrin.h>
#define JG_HAVE_SSE2 1
#else
#define JG_HAVE_SSE2 0
#endif
#include <cstdlib>
#include <cstring>

namespace jg {

namespace {
constexpr uint8_t M_SOF0 = 0xC0, M_SOF1 = 0xC1, M_DHT = 0xC4, M_RST0 = 0xD0, M_RST7 = 0xD7,
                  M_SOI = 0xD8, M_EOI = 0xD9, M_SOS = 0xDA, M_DQT = 0xDB, M_DRI = 0xDD;
constexpr int kNatural[64] = JG_ORDER_NATURAL;

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
} // namespace

#if JG_HAVE_SSE2
/// Result of scanning entropy-coded bytes [from, lim): the first FF that is not followed by 00 (a
/// marker or fill byte) if there is one, and the number of FF bytes before it (all stuffed).
struct WindowScan {
    const uint8_t* marker;
    uint32_t ff;
};

/// 64-byte blocks aligned to `grid`; `lim - grid` is a multiple of 64 and byte `lim` is readable.
#define JG_SCAN_WINDOW_BODY(MASKS)                                                                  \
    const size_t fo    = static_cast<size_t>(from - grid);                                          \
    const uint8_t* blk = grid + (fo & ~static_cast<size_t>(63));                                    \
    uint64_t keep      = ~0ull << (fo & 63);                                                        \
    uint32_t count     = 0;                                                                         \
    for (; blk < lim; blk += 64, keep = ~0ull) {                                                    \
        uint64_t ff, zr;                                                                            \
        MASKS;                                                                                      \
        ff &= keep;                                                                                 \
        /* no early-out for blocks without FF: with an FF in every second block (noisy images) that    \
           branch mispredicts half the time and costs more than the few instructions it skips */      \
        const uint64_t next_zero = (zr >> 1) | (static_cast<uint64_t>(blk[64] == 0) << 63);         \
        const uint64_t mk        = ff & ~next_zero;                                                 \
        if (__builtin_expect(mk != 0, 0)) {                                                         \
            const int k = __builtin_ctzll(mk);                                                      \
            count += static_cast<uint32_t>(__builtin_popcountll(ff & ((1ull << k) - 1ull)));        \
            return WindowScan{blk + k, count};                                                      \
        }                                                                                           \
        count += static_cast<uint32_t>(__builtin_popcountll(ff));                                   \
    }                                                                                               \
    return WindowScan{nullptr, count};

WindowScan scan_window_sse2(const uint8_t* from, const uint8_t* lim, const uint8_t* grid)
{
    JG_SCAN_WINDOW_BODY(
        ff = 0; zr = 0; for (int k = 0; k < 4; ++k) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(blk + 16 * k));
            ff |= static_cast<uint64_t>(static_cast<uint32_t>(_mm_movemask_epi8(_mm_cmpeq_epi8(a, _mm_set1_epi8(static_cast<char>(0xFF)))))) << (16 * k);
            zr |= static_cast<uint64_t>(static_cast<uint32_t>(_mm_movemask_epi8(_mm_cmpeq_epi8(a, _mm_setzero_si128())))) << (16 * k);
        })
}

__attribute__((target("avx2,popcnt"))) WindowScan scan_window_avx2(const uint8_t* from, const uint8_t* lim, const uint8_t* grid)
{
    JG_SCAN_WINDOW_BODY(
        const __m256i a0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(blk));
        const __m256i a1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(blk + 32));
        const __m256i vf = _mm256_set1_epi8(static_cast<char>(0xFF));
        const __m256i vz = _mm256_setzero_si256();
        ff = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(a0, vf))) |
             static_cast<uint64_t>(static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(a1, vf)))) << 32;
        zr = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(a0, vz))) |
             static_cast<uint64_t>(static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(a1, vz)))) << 32;)
}
__attribute__((target("avx512f,avx512bw,popcnt"))) WindowScan scan_window_avx512(const uint8_t* from, const uint8_t* lim, const uint8_t* grid)
{
    JG_SCAN_WINDOW_BODY(
        const __m512i a = _mm512_loadu_si512(reinterpret_cast<const void*>(blk));
        ff = _mm512_cmpeq_epi8_mask(a, _mm512_set1_epi8(static_cast<char>(0xFF)));
        zr = _mm512_testn_epi8_mask(a, a);)
}
#undef JG_SCAN_WINDOW_BODY
#endif

void build_huff_table(
    std::vector<uint8_t>& t, const uint8_t (&num_codes)[16], const uint8_t* huffval, int count, bool is_dc)
{
    const int lb        = is_dc ? kLutBitsDc : kLutBitsAc;
    const size_t lut_sz = static_cast<size_t>(2) << lb;
    t.assign(lut_sz + kHuffAuxSize, 0);
    uint16_t* lut  = reinterpret_cast<uint16_t*>(t.data());
    uint16_t* lim  = reinterpret_cast<uint16_t*>(t.data() + lut_sz);
    uint16_t* voff = lim + 8;
    uint8_t* hv    = t.data() + lut_sz + 32;
    for (int i = 0; i < count && i < 256; ++i) hv[i] = huffval[i];
    // canonical code assignment, T.81 Annex C; lut entry 0 = "longer than LB bits or undefined"
    int idx       = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 16; ++l) {
        const int first_idx = idx;
        const uint32_t first_code = code;
        for (int i = 0; i < num_codes[l - 1] && idx < 256; ++i) {
            if (l <= lb) {
                const int shift  = lb - l;
                const uint32_t e = huff_entry(l, hv[idx], is_dc);
                const uint32_t lo = (code << shift) & ((1u << lb) - 1);
                for (uint32_t j = 0; j < (1u << shift); ++j) lut[(lo + j) & ((1u << lb) - 1)] = static_cast<uint16_t>(e);
            }
            ++idx;
            ++code;
        }
        if (l >= 9) {
            // `code` now counts every code of length <= l; left-aligned it separates them from longer ones
            const uint32_t left = code << (16 - l);
            lim[l - 9]  = static_cast<uint16_t>(left > 0xFFFFu ? 0xFFFFu : left);
            voff[l - 9] = num_codes[l - 1] ? static_cast<uint16_t>((first_idx - static_cast<int>(first_code)) & 0xFF) : 0;
        }
        code <<= 1;
    }
    // Second-level tables for the LB-bit prefixes that start a valid longer code. Their entries are
    // what the long-code path returns for each completion, so the two paths cannot disagree.
    uint32_t prefix_of[256];
    int num_sub = 0;
    idx         = 0;
    code        = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < num_codes[l - 1] && idx < 256; ++i, ++idx, ++code) {
            if (l <= lb) continue;
            const uint32_t prefix = (code >> (l - lb)) & ((1u << lb) - 1);
            if (lut[prefix] != 0 || num_sub == kMaxSubTables) continue; // shorter code / already has one / budget
            prefix_of[num_sub++] = prefix;
            lut[prefix]          = static_cast<uint16_t>(num_sub << 5);
        }
        code <<= 1;
    }
    if (num_sub == 0) return;
    t.resize(lut_sz + kHuffAuxSize + static_cast<size_t>(num_sub) * kSubTableSize, 0);
    const uint8_t* aux = t.data() + lut_sz;
    uint16_t* sub      = reinterpret_cast<uint16_t*>(t.data() + lut_sz + kHuffAuxSize);
    const int known    = lb + kSubBits;          // window bits a second-level index pins down
    for (int k = 0; k < num_sub; ++k) {
        for (uint32_t i2 = 0; i2 < (1u << kSubBits); ++i2) {
            const uint32_t v    = known >= 16 ? ((prefix_of[k] << kSubBits) | i2) >> (known - 16)
                                              : ((prefix_of[k] << kSubBits) | i2) << (16 - known);
            const uint32_t free = known >= 16 ? 0u : (1u << (16 - known)) - 1u;
            // the entry must not depend on the window bits the index does not cover
            const uint32_t e0 = huff_long_code(aux, v << 16, is_dc);
            const uint32_t e1 = huff_long_code(aux, (v | free) << 16, is_dc);
            const int codelen = static_cast<int>(e0 & 31u) - static_cast<int>((e0 >> 5) & 15u);
            sub[k * (1 << kSubBits) + i2] = static_cast<uint16_t>((e0 == e1 && codelen <= known) ? e0 : 0u);
        }
    }
}

void widen_huff_table(const std::vector<uint8_t>& t, bool is_dc, std::vector<uint8_t>& out)
{
    const int lb          = is_dc ? kLutBitsDc : kLutBitsAc;
    const uint32_t n      = 1u << lb;
    const uint16_t* lut   = reinterpret_cast<const uint16_t*>(t.data());
    const size_t rest     = t.size() - 2u * n;
    out.assign(static_cast<size_t>(kSyncEntryBytes) * n + rest, 0);
    uint32_t* wide = reinterpret_cast<uint32_t*>(out.data());
    for (uint32_t idx = 0; idx < n; ++idx) {
        const uint32_t single = lut[idx];
        uint32_t multi        = single; // no second symbol: the high half repeats the low one
        if (!is_dc && (single & 31u) != 0) {
            // Decode greedily inside the lb index bits: a symbol counts if its code AND its magnitude bits lie
            // inside them (its first-level entry then does not depend on the bits behind the index).
            uint32_t bits = 0, pre = 0, total_adv = 0;
            int count = 0;
            while (true) {
                const uint32_t e   = lut[(idx << bits) & (n - 1u)];
                const uint32_t len = e & 31u, adv = e >> 9;
                if (len == 0 || bits + len > static_cast<uint32_t>(lb)) break;
                if (count > 0 && total_adv > static_cast<uint32_t>(kMultiMaxPre)) break; // the earlier symbols' advance has four bits
                pre = total_adv;
                total_adv += adv;
                bits += len;
                ++count;
                if (adv == kEobAdvance) break; // end of block: whatever follows belongs to the next data unit
            }
            if (count >= 2) multi = bits | pre << 5 | total_adv << 9;
        }
        wide[idx] = single | multi << 16;
    }
    std::memcpy(out.data() + static_cast<size_t>(kSyncEntryBytes) * n, t.data() + 2u * n, rest);
}

jpeggpu_status Reader::read_sof(const Logger& log)
{
    if (remaining() < 2) return JPEGGPU_INVALID_JPEG;
    const uint16_t length = u16();
    if (length < 2) return JPEGGPU_INVALID_JPEG;
    if (remaining() < static_cast<size_t>(length - 2)) return JPEGGPU_INCOMPLETE_BITSTREAM;
    if (length < 8) return JPEGGPU_INVALID_JPEG;
    const uint8_t precision = u8();
    if (precision != 8) {
        log.log("\tunsupported sample precision %d, only 8 is supported\n", precision);
        return JPEGGPU_NOT_SUPPORTED;
    }
    const uint16_t lines   = u16();
    const uint16_t samples = u16();
    if (lines == 0 || samples == 0) {
        log.log("\tinvalid size x=%d, y=%d\n", samples, lines);
        return JPEGGPU_INVALID_JPEG;
    }
    s.size_x         = samples;
    s.size_y         = lines;
    const uint8_t nc = u8();
    if (nc == 0) return JPEGGPU_INVALID_JPEG;
    if (nc > kMaxComp) {
        log.log("\ttoo many components %d\n", nc);
        return JPEGGPU_NOT_SUPPORTED;
    }
    if (length != 8 + 3 * nc) return JPEGGPU_INVALID_JPEG;
    s.num_comp = nc;
    log.log("\tsize_x: %d, size_y: %d, num_components: %d\n", s.size_x, s.size_y, s.num_comp);
    s.hs_max = s.vs_max = 0;
    for (int c = 0; c < nc; ++c) {
        Component& comp  = s.comp[c];
        comp.id          = u8();
        const uint8_t sf = u8();
        const int hs = sf >> 4, vs = sf & 15;
        if (hs < 1 || hs > 4 || vs < 1 || vs > 4) {
            log.log("\tinvalid sampling factor (%d, %d)\n", hs, vs);
            return JPEGGPU_INVALID_JPEG;
        }
        // a single-component frame is decoded as 1x1 whatever the header says (reference :147-153)
        comp.hs         = nc == 1 ? 1 : hs;
        comp.vs         = nc == 1 ? 1 : vs;
        const uint8_t q = u8();
        if (q > 3) {
            log.log("\tinvalid quantization table index (%d)\n", q);
            return JPEGGPU_INVALID_JPEG;
        }
        comp.qidx = q;
        for (int d = 0; d < c; ++d) {
            if (s.comp[d].id == comp.id) return JPEGGPU_INVALID_JPEG;
        }
        log.log("\tc_id: %d, ssx: %d, ssy: %d, qi: %d\n", comp.id, comp.hs, comp.vs, comp.qidx);
        s.hs_max = std::max(s.hs_max, comp.hs);
        s.vs_max = std::max(s.vs_max, comp.vs);
    }
    for (int c = 0; c < nc; ++c) {
        Component& comp = s.comp[c];
        comp.size_x     = ceil_div(s.size_x * comp.hs, s.hs_max); // T.81 A.1.1
        comp.size_y     = ceil_div(s.size_y * comp.vs, s.vs_max);
    }
    return JPEGGPU_SUCCESS;
}

jpeggpu_status Reader::read_dht(const Logger& log)
{
    if (remaining() < 2) return JPEGGPU_INVALID_JPEG;
    const uint16_t length = u16();
    if (length < 2 || remaining() < static_cast<size_t>(length - 2)) {
        log.log("\ttoo few bytes in DHT segment\n");
        return JPEGGPU_INVALID_JPEG;
    }
    int rem = length - 2;
    while (rem > 0) {
        const uint8_t index = u8();
        --rem;
        const int tc = index >> 4, th = index & 15;
        if (tc != 0 && tc != 1) {
            log.log("\tinvalid Huffman table class\n");
            return JPEGGPU_INVALID_JPEG;
        }
        if (th > 3) {
            log.log("\tHuffman table index must be 0, 1, 2, or 3\n");
            return JPEGGPU_NOT_SUPPORTED;
        }
        if (rem < 16) return JPEGGPU_INVALID_JPEG;
        uint8_t num_codes[16];
        int count = 0;
        for (int i = 0; i < 16; ++i) {
            num_codes[i] = u8();
            count += num_codes[i];
        }
        rem -= 16;
        if (count > 256 || rem < count) {
            log.log("\tinvalid value count in DHT segment\n");
            return JPEGGPU_INVALID_JPEG;
        }
        // reject over-subscribed code lengths: the canonical code would not fit its length
        uint32_t code = 0;
        for (int l = 0; l < 16; ++l) {
            code += num_codes[l];
            if (code > (1u << (l + 1))) return JPEGGPU_INVALID_JPEG;
            code <<= 1;
        }
        log.log("\t%s Huffman table index %d\n", tc == 0 ? "DC" : "AC", th);
        // A decoder parses image after image, nearly always with the same tables: the device forms are rebuilt only
        // when the DHT payload (16 counts + the values) differs from the one they were built from.
        std::vector<uint8_t>& key = dht_key_[tc][th];
        const uint8_t* payload    = cur_ - 16;
        const size_t payload_len  = 16u + static_cast<size_t>(count);
        const bool same = key.size() == payload_len && std::memcmp(key.data(), payload, payload_len) == 0;
        if (!same) {
            std::vector<uint8_t>& tab = tc == 0 ? dc_tab_[th] : ac_tab_[th];
            build_huff_table(tab, num_codes, cur_, count, tc == 0);
            widen_huff_table(tab, tc == 0, tc == 0 ? dc_tab_sync_[th] : ac_tab_sync_[th]);
            key.assign(payload, payload + payload_len);
        }
        (tc == 0 ? dc_defined_ : ac_defined_)[th] = true;
        cur_ += count;
        rem -= count;
    }
    return JPEGGPU_SUCCESS;
}

jpeggpu_status Reader::read_dqt(const Logger& log)
{
    if (remaining() < 2) return JPEGGPU_INVALID_JPEG;
    const uint16_t length = u16();
    if (length < 2 || remaining() < static_cast<size_t>(length - 2)) {
        log.log("\ttoo few bytes in DQT segment\n");
        return JPEGGPU_INVALID_JPEG;
    }
    int rem = length - 2;
    while (rem > 0) {
        const uint8_t info = u8();
        --rem;
        const int precision = info >> 4, id = info & 15;
        if ((precision != 0 && precision != 1) || id > 3) {
            log.log("\tinvalid precision or id value\n");
            return JPEGGPU_INVALID_JPEG;
        }
        // 16-bit entries (Pq = 1): the reference returns NOT_SUPPORTED (src/reader.cpp:517-520); libjpeg
        // writes such tables at low quality without -baseline, so they are read
        const int bytes = precision ? 128 : 64;
        if (rem < bytes) return JPEGGPU_INVALID_JPEG;
        // A table that a component of an earlier scan dequantises with must not be replaced: all
        // scans are dequantised together at the end (reference src/reader.cpp:524-541 intends this).
        bool in_use = false;
        for (int c = 0; c < s.num_comp; ++c) in_use |= comp_in_scan_[c] && s.comp[c].qidx == id;
        for (int j = 0; j < 64; ++j) {
            const uint16_t q = precision ? u16() : u8();
            if (!in_use) s.qtable[id][kNatural[j]] = q;
        }
        qt_defined_[id] = true;
        rem -= bytes;
    }
    return JPEGGPU_SUCCESS;
}

jpeggpu_status Reader::read_dri(const Logger& log)
{
    if (remaining() < 2) return JPEGGPU_INVALID_JPEG;
    const uint16_t length = u16();
    if (length != 4 || remaining() < 2) return JPEGGPU_INVALID_JPEG;
    const uint16_t rsti = u16();
    if (s.num_scans > 0 && s.restart_interval != rsti) {
        log.log("\tredefined restart interval\n");
        return JPEGGPU_NOT_SUPPORTED;
    }
    s.restart_interval = rsti;
    log.log("\trestart_interval: %d\n", s.restart_interval);
    return JPEGGPU_SUCCESS;
}

jpeggpu_status Reader::skip_segment(const Logger& log)
{
    if (remaining() < 2) return JPEGGPU_INVALID_JPEG;
    const uint16_t length = u16();
    if (length < 2) return JPEGGPU_INVALID_JPEG;
    if (remaining() < static_cast<size_t>(length - 2)) return JPEGGPU_INCOMPLETE_BITSTREAM;
    log.log("\twarning: skipping this segment\n");
    cur_ += length - 2;
    return JPEGGPU_SUCCESS;
}

jpeggpu_status Reader::read_sos(const Logger& log)
{
    if (!found_sof_) return JPEGGPU_INVALID_JPEG;
    if (remaining() < 3) {
        log.log("\ttoo few bytes in SOS segment\n");
        return JPEGGPU_INVALID_JPEG;
    }
    const uint16_t length = u16();
    if (length < 3) return JPEGGPU_INVALID_JPEG;
    const uint8_t ns = u8();
    if (ns < 1 || ns > 4) {
        log.log("\tinvalid number of components in scan %d\n", ns);
        return JPEGGPU_INVALID_JPEG;
    }
    if (s.num_scans >= kMaxScans) return JPEGGPU_INVALID_JPEG;
    if (length != 6 + 2 * ns) return JPEGGPU_INVALID_JPEG;
    if (remaining() < static_cast<size_t>(2 * ns + 3)) return JPEGGPU_INCOMPLETE_BITSTREAM;

    Scan& scan            = s.scans[s.num_scans];
    scan                  = Scan{};
    scan.num_comp         = ns;
    const bool interleave = ns > 1;
    scan.du_per_mcu       = 0;
    for (int a = 0; a < ns; ++a) {
        ScanComponent& sc = scan.comp[a];
        const uint8_t sel = u8();
        const uint8_t tab = u8();
        const int id_dc = tab >> 4, id_ac = tab & 15;
        log.log("\tc_id: %d, dc: %d, ac: %d\n", sel, id_dc, id_ac);
        int ci = -1;
        for (int i = 0; i < s.num_comp; ++i) {
            if (s.comp[i].id == sel) {
                ci = i;
                break;
            }
        }
        if (ci < 0) {
            log.log("\tinvalid component selector\n");
            return JPEGGPU_INVALID_JPEG;
        }
        // T.81 A.2: scan components follow frame order; a component is coded in exactly one scan
        if (a > 0 && ci <= scan.comp[a - 1].comp_idx) return JPEGGPU_INVALID_JPEG;
        if (comp_in_scan_[ci]) return JPEGGPU_INVALID_JPEG;
        if (id_dc > 3 || id_ac > 3) return JPEGGPU_INVALID_JPEG;
        if (!dc_defined_[id_dc] || !ac_defined_[id_ac]) return JPEGGPU_INVALID_JPEG;
        const Component& comp = s.comp[ci];
        if (!qt_defined_[comp.qidx]) {
            log.log("\tquantization table at index %d not defined\n", comp.qidx);
            return JPEGGPU_INVALID_JPEG;
        }
        sc.comp_idx = ci;
        sc.dc_id    = id_dc;
        sc.ac_id    = id_ac;
        sc.h        = interleave ? comp.hs : 1;
        sc.v        = interleave ? comp.vs : 1;
        sc.data_x   = ceil_div(comp.size_x, 8 * sc.h) * 8 * sc.h; // T.81 A.2.4
        sc.data_y   = ceil_div(comp.size_y, 8 * sc.v) * 8 * sc.v;
        const int mx = sc.data_x / (8 * sc.h), my = sc.data_y / (8 * sc.v);
        if (a > 0 && (mx != scan.mcus_x || my != scan.mcus_y)) {
            // sampling factors that do not divide the maximum can disagree on the MCU count
            log.log("\tcomponents disagree on the number of MCUs\n");
            return JPEGGPU_NOT_SUPPORTED;
        }
        scan.mcus_x = mx;
        scan.mcus_y = my;
        scan.du_per_mcu += sc.h * sc.v;
    }
    if (scan.du_per_mcu > kMaxDuPerMcu) {
        log.log("\ttoo many data units in mcu\n");
        return JPEGGPU_INVALID_JPEG;
    }
    for (int a = 0; a < ns; ++a) comp_in_scan_[scan.comp[a].comp_idx] = true;
    u8(); // spectral selection start (0 for baseline)
    u8(); // spectral selection end (63)
    u8(); // successive approximation (0)

    // pack the tables in force for this scan, once in each form (jg_defs.h); components that select the same
    // table share it; the cursor ring sits behind the tables: one entry per data unit of the MCU
    const auto build_pack = [&](const std::vector<uint8_t> (&dc_tabs)[4], const std::vector<uint8_t> (&ac_tabs)[4],
                                std::vector<uint8_t>& pack, uint32_t& cursor_off, uint32_t limit, bool write_pack) -> bool {
        uint32_t dc_off[kMaxComp], ac_off[kMaxComp];
        pack.clear();
        for (int a = 0; a < ns; ++a) {
            const ScanComponent& sc = scan.comp[a];
            int same_dc = -1, same_ac = -1;
            for (int b = 0; b < a; ++b) {
                if (scan.comp[b].dc_id == sc.dc_id) same_dc = b;
                if (scan.comp[b].ac_id == sc.ac_id) same_ac = b;
            }
            if (same_dc >= 0) {
                dc_off[a] = dc_off[same_dc];
            } else {
                dc_off[a] = static_cast<uint32_t>(pack.size());
                pack.insert(pack.end(), dc_tabs[sc.dc_id].begin(), dc_tabs[sc.dc_id].end());
            }
            if (same_ac >= 0) {
                ac_off[a] = ac_off[same_ac];
            } else {
                ac_off[a] = static_cast<uint32_t>(pack.size());
                pack.insert(pack.end(), ac_tabs[sc.ac_id].begin(), ac_tabs[sc.ac_id].end());
                if (write_pack) {
                    // first-level AC entries the write pass's ordinary step cannot take carry kEntrySlow (jg_defs.h): no
                    // length (second level or long code), or a category with an escape entry
                    uint16_t* lut = reinterpret_cast<uint16_t*>(pack.data() + ac_off[a]);
                    for (uint32_t i = 0; i < (1u << kLutBitsAc); ++i)
                        if ((lut[i] & 31u) == 0 || ((lut[i] >> 5) & 15u) >= static_cast<uint32_t>(kEscapeFromCategory)) lut[i] |= static_cast<uint16_t>(kEntrySlow);
                }
            }
        }
        // 16-bit offsets in the ring, 16-bit LDS addresses on the device: cannot trip while the static_assert on
        // the pack sizes in jg_defs.h holds, and must fail rather than wrap if a constant is ever raised without it
        if (pack.size() + static_cast<size_t>(kMaxDuPerMcu) * sizeof(CursorEntry) > limit) return false;
        cursor_off = static_cast<uint32_t>(pack.size());
        std::vector<CursorEntry> ring;
        for (int a = 0; a < ns; ++a) {
            for (int k = 0; k < scan.comp[a].h * scan.comp[a].v; ++k) {
                CursorEntry ce;
                const uint32_t du = static_cast<uint32_t>(ring.size());
                ce.tabs = dc_off[a] | ac_off[a] << 16;
                ce.meta = 16u * a | du << 8;
                ce.self = cursor_off + 16u * du;
                ce.next = cursor_off + 16u * (static_cast<int>(du) + 1 == scan.du_per_mcu ? 0u : du + 1u);
                ring.push_back(ce);
            }
        }
        const uint8_t* rb = reinterpret_cast<const uint8_t*>(ring.data());
        pack.insert(pack.end(), rb, rb + ring.size() * sizeof(CursorEntry));
        return true;
    };
    if (!build_pack(dc_tab_, ac_tab_, scan.table_pack, scan.cursor_off, kMaxTablePack, true) ||
        !build_pack(dc_tab_sync_, ac_tab_sync_, scan.table_pack_sync, scan.cursor_off_sync, kMaxTablePackSync, false))
        return JPEGGPU_INTERNAL_ERROR;
    const int total_mcus  = scan.mcus_x * scan.mcus_y;
    scan.mcus_per_segment = s.restart_interval ? s.restart_interval : total_mcus;
    scan.num_du           = total_mcus * scan.du_per_mcu;
    scan.begin            = static_cast<size_t>(cur_ - base_);
    if (s.num_scans == 0 && subseq_request_ <= 0) {
        // per-image choice (jg_reader.hpp): what is left of the file bounds the scan, DRI and the geometry give the segments
        subseq_bytes_ = choose_subseq_bytes(-subseq_request_, static_cast<size_t>(end_ - cur_),
                                            static_cast<size_t>(ceil_div(total_mcus, scan.mcus_per_segment)));
        log.log("\tsubsequence size chosen for this image: %d bytes\n", subseq_bytes_);
    }
    if (s.num_scans == 0) s.xfer_begin = (scan.begin - 1) & ~static_cast<size_t>(15);
    ++s.num_scans;
    bool last_scan = true; // this scan completes the frame's components (comp_in_scan_ already counts its own)
    for (int c = 0; c < s.num_comp; ++c) last_scan = last_scan && comp_in_scan_[c];
    if (device_scan_ && last_scan) {
        // Device-side front end: everything up to the end of the file is transferred, the device finds the
        // restart markers and the end of the scan. Upper bounds: a segment of d data bytes has
        // ceil(d / subsequence) subsequences, so at most (bytes / subsequence) + segments in total; a chunk
        // is a 4 KiB window of one segment.
        // The copy need not run to the end of the file: nothing behind the LAST end-of-image marker can belong to
        // the scan. Found from the back -- free for a file that ends in FF D9, a sweep over the padding for one
        // that carries zeros or a trailer behind it (reference photo: 1.17 MB of zeros) -- without touching the scan.
        const uint8_t* stop = end_;
        for (const uint8_t* q = end_; q - (base_ + scan.begin) >= 2;) {
            q = static_cast<const uint8_t*>(memrchr(base_ + scan.begin + 1, 0xD9, static_cast<size_t>(q - (base_ + scan.begin + 1))));
            if (!q) break;
            if (q[-1] == 0xFF) {
                stop = q + 1;
                break;
            }
        }
        const size_t file_end = static_cast<size_t>(stop - base_);
        const size_t bytes    = file_end - scan.begin;
        // In a file of several scans the device takes the last one only if that is most of the work: its front end costs
        // four launches and its stages launch apart from the host-walked scans'. BASELINE configs[3] (three scans, the
        // last one 19 % of the bytes): host parse 0.105 -> 0.099 ms, but p50 0.58 -> 0.76 ms with it (round 4).
        size_t walked = 0;
        for (int k = 0; k + 1 < s.num_scans; ++k) walked += s.scans[k].end - s.scans[k].begin;
        const bool worth = bytes >= walked;
        const size_t segments = static_cast<size_t>(ceil_div(total_mcus, scan.mcus_per_segment));
        // the device looks at the windows from the one that holds this scan's first byte on (earlier scans' bytes lie in
        // front of it: they were walked on the host)
        const size_t win0     = (scan.begin - s.xfer_begin) / kDestuffWin;
        const size_t windows  = (file_end - s.xfer_begin + kDestuffWin - 1) / kDestuffWin - win0;
        const size_t subseq   = bytes / static_cast<size_t>(subseq_bytes_) + segments + 1;
        if (worth && segments <= (1u << 20) && subseq < (1u << 24) && bytes < (1u << 27)) {
            scan.device_walk     = true;
            scan.front_win0      = static_cast<uint32_t>(win0);
            scan.expect_segments = static_cast<int>(segments);
            scan.num_subseq      = static_cast<int>(subseq);
            scan.max_chunks      = static_cast<int>(windows + segments + 1);
            scan.max_tail_parts  = static_cast<int>(subseq / kTailPartSubseq + 3);
            scan.end             = file_end;
            s.xfer_end           = scan.end;
            stop_                = true;
            return JPEGGPU_SUCCESS;
        }
    }
    return walk_scan(scan, log);
}

/// Walk the entropy-coded bytes of one scan (reference src/reader.cpp:447-489): find restart
/// markers and the terminating marker, count destuffed bytes per segment, and emit the destuff
/// work list with precomputed destination offsets.
jpeggpu_status Reader::walk_scan(Scan& scan, const Logger& log)
{
    const size_t xb = s.xfer_begin;
    const auto boff = [&](const uint8_t* p) { return static_cast<size_t>(p - base_) - xb; };

    const uint8_t* seg_begin = cur_;
    size_t chunk_begin       = boff(cur_);
    size_t next_win          = (chunk_begin / kDestuffWin + 1) * kDestuffWin;
    uint32_t ff_in_chunk     = 0;
    size_t seg_first_chunk   = scan.chunks.size();
    int seg_dst_base         = 0; // in bytes of the destuffed buffer
    size_t chunk_dst         = 0;
    const uint8_t* pos       = cur_;
    (void)seg_begin;

    const auto emit = [&](size_t b, size_t e) {
        DestuffChunk c;
        c.win_off = static_cast<uint32_t>(b / kDestuffWin * kDestuffWin);
        c.begin   = static_cast<uint32_t>(b);
        c.end     = static_cast<uint32_t>(e);
        c.dst_off = static_cast<uint32_t>(chunk_dst);
        c.pad_end = 0;
        c.seg     = static_cast<int32_t>(scan.segments.size());
        c.first   = scan.chunks.size() == seg_first_chunk ? 1u : 0u;
        c.reserved = 0;
        scan.chunks.push_back(c);
        chunk_dst += (e - b) - ff_in_chunk;
        ff_in_chunk = 0;
    };

    // Bulk part of the walk (scan_window above): whole destuff windows at a time; the per-byte code
    // below runs at real markers and near the end of the buffer.
    const uint8_t* const grid = base_ + xb;
    const auto skip_data = [&](const uint8_t* from) -> const uint8_t* {
#if JG_HAVE_SSE2
        // JPEGGPU_HOST_SIMD = scalar | sse2 | avx2 pins the path (tests cover all four); default: best available
        const char* pin      = std::getenv("JPEGGPU_HOST_SIMD");
        if (pin && std::strcmp(pin, "scalar") == 0) return from;
        const bool have_avx2 = !(pin && std::strcmp(pin, "sse2") == 0) && __builtin_cpu_supports("avx2") &&
                               __builtin_cpu_supports("popcnt");
        const bool have_avx512 = have_avx2 && !(pin && std::strcmp(pin, "avx2") == 0) && __builtin_cpu_supports("avx512bw") &&
                                 __builtin_cpu_supports("avx512f");
        // blocks [.., last_blk) can be scanned: the byte behind a block must be readable
        const uint8_t* const last_blk = end_ - grid > 64 ? grid + ((static_cast<size_t>(end_ - grid) - 1) & ~static_cast<size_t>(63)) : grid;
        while (true) {
            const uint8_t* wend = grid + next_win;
            if (from >= wend) {
                emit(chunk_begin, next_win);
                chunk_begin = next_win;
                next_win += kDestuffWin;
                continue;
            }
            const uint8_t* lim = wend < last_blk ? wend : last_blk;
            if (from >= lim) return from;
            const WindowScan r = have_avx512 ? scan_window_avx512(from, lim, grid)
                                 : have_avx2 ? scan_window_avx2(from, lim, grid)
                                             : scan_window_sse2(from, lim, grid);
            ff_in_chunk += r.ff;
            if (r.marker) return r.marker;
            from = lim;
            if (lim != wend) return from;
        }
#else
        return from;
#endif
    };

    while (true) {
        pos = skip_data(pos);
        const uint8_t* q =
            static_cast<const uint8_t*>(std::memchr(pos, 0xFF, static_cast<size_t>(end_ - pos)));
        if (q == nullptr || q + 1 >= end_) return JPEGGPU_INVALID_JPEG; // no end-of-image marker
        const size_t qo = boff(q);
        while (next_win <= qo) {
            emit(chunk_begin, next_win);
            chunk_begin = next_win;
            next_win += kDestuffWin;
        }
        uint8_t m = q[1];
        if (m == 0) { // stuffed byte: FF 00 stands for a data byte FF
            ++ff_in_chunk;
            pos = q + 2;
            continue;
        }
        const uint8_t* mk = q; // skip fill bytes (any number of FF before the marker code)
        while (m == 0xFF) {
            ++mk;
            if (mk + 1 >= end_) return JPEGGPU_INVALID_JPEG;
            m = mk[1];
        }
        if (m == 0) return JPEGGPU_INVALID_JPEG; // FF FF 00 is not a legal sequence
        // the segment's data ends at q
        if (qo > chunk_begin || scan.chunks.size() == seg_first_chunk) emit(chunk_begin, qo);
        const size_t seg_bytes = chunk_dst - static_cast<size_t>(seg_dst_base);
        if (seg_bytes > (1u << 27)) return JPEGGPU_NOT_SUPPORTED; // bit positions are 32-bit
        // A restart interval holds at least one MCU, so at least one byte: two markers back to back (which the
        // reference's walk, src/reader.cpp:447-489, takes for a segment of no subsequences) are not a JPEG. Kernels that
        // work segment by segment would otherwise meet a segment without a first subsequence.
        if (seg_bytes == 0) {
            log.log("\trestart segment %d holds no entropy-coded data\n", static_cast<int>(scan.segments.size()));
            return JPEGGPU_INVALID_JPEG;
        }
        Segment seg;
        seg.subseq_offset = scan.num_subseq;
        seg.subseq_count  = static_cast<int>((seg_bytes + subseq_bytes_ - 1) / subseq_bytes_);
        scan.num_subseq += seg.subseq_count;
        const size_t padded_end = static_cast<size_t>(scan.num_subseq) * subseq_bytes_;
        scan.chunks.back().pad_end = static_cast<uint32_t>(padded_end);
        scan.segments.push_back(seg);
        if (padded_end > (1u << 31)) return JPEGGPU_NOT_SUPPORTED;

        const bool is_rst = M_RST0 <= m && m <= M_RST7;
        if (!is_rst) {
            cur_     = mk; // the marker loop continues at the FF of the terminating marker
            scan.end = static_cast<size_t>(mk - base_);
            break;
        }
        pos = mk + 2;
        if (pos >= end_) return JPEGGPU_INVALID_JPEG;
        // next segment
        seg_dst_base    = static_cast<int>(padded_end);
        chunk_dst       = padded_end;
        chunk_begin     = boff(pos);
        next_win        = (chunk_begin / kDestuffWin + 1) * kDestuffWin;
        ff_in_chunk     = 0;
        seg_first_chunk = scan.chunks.size();
    }

    const int total_mcus = scan.mcus_x * scan.mcus_y;
    const int expect     = ceil_div(total_mcus, scan.mcus_per_segment);
    if (static_cast<int>(scan.segments.size()) != expect) {
        log.log(
            "\tscan has %d restart segments, geometry requires %d\n",
            static_cast<int>(scan.segments.size()),
            expect);
        return JPEGGPU_INVALID_JPEG;
    }
    s.xfer_end = scan.end;
    // Flows never cross a segment, so runs of whole segments can be synchronised by independent
    // workgroups: cut the scan at segment starts into parts of about kTailPartSubseq subsequences.
    scan.tail_parts.clear();
    scan.tail_parts.push_back(0);
    const int part = kTailPartSubseq;
    for (const Segment& seg : scan.segments) {
        if (seg.subseq_offset - scan.tail_parts.back() >= part) scan.tail_parts.push_back(seg.subseq_offset);
    }
    scan.tail_parts.push_back(scan.num_subseq);
    return JPEGGPU_SUCCESS;
}

/// Keep only this decoder's share of the scan's restart segments (segments are independent for the
/// Huffman decode and for DC prediction, reference src/decode_dc.cu:119-144): segments [rank * n / world,
/// (rank + 1) * n / world), renumbered from 0, with the destuff work list, the transferred byte range and the tail
/// parts cut to them. Needs a single scan whose restart interval is a whole number of MCU rows, so that a run of
/// segments is a horizontal band of every plane.
jpeggpu_status Reader::apply_segment_shard(int rank, int world, const Logger& log)
{
    if (s.num_scans != 1 || s.restart_interval == 0) {
        log.log("\tsegment shard needs one scan with restart markers\n");
        return JPEGGPU_NOT_SUPPORTED;
    }
    Scan& scan = s.scans[0];
    if (scan.mcus_per_segment % scan.mcus_x != 0) {
        log.log("\tsegment shard needs a restart interval of whole MCU rows\n");
        return JPEGGPU_NOT_SUPPORTED;
    }
    const int n = static_cast<int>(scan.segments.size());
    const int a = static_cast<int>(static_cast<long long>(n) * rank / world);
    const int b = static_cast<int>(static_cast<long long>(n) * (rank + 1) / world);
    const int total_mcus = scan.mcus_x * scan.mcus_y;
    scan.first_segment   = a;
    scan.total_segments  = n;
    scan.first_mcu       = a * scan.mcus_per_segment;
    scan.shard_mcus      = std::min(b * scan.mcus_per_segment, total_mcus) - std::min(a * scan.mcus_per_segment, total_mcus);
    scan.num_du          = scan.shard_mcus * scan.du_per_mcu;
    if (a == b) { // more ranks than segments: nothing to do on this one
        scan.segments.clear();
        scan.chunks.clear();
        scan.tail_parts.assign(1, 0);
        scan.num_subseq = 0;
        s.xfer_end      = s.xfer_begin;
        return JPEGGPU_SUCCESS;
    }
    const uint32_t first_sub = static_cast<uint32_t>(scan.segments[a].subseq_offset);
    const uint32_t dst_shift = first_sub * static_cast<uint32_t>(subseq_bytes_);
    std::vector<Segment> segs(scan.segments.begin() + a, scan.segments.begin() + b);
    for (Segment& g : segs) g.subseq_offset -= static_cast<int>(first_sub);
    std::vector<DestuffChunk> chunks;
    for (const DestuffChunk& c : scan.chunks)
        if (c.seg >= a && c.seg < b) chunks.push_back(c);
    // the window grid of the destuff kernel starts at offset 0 of the transferred bytes: shift by whole windows
    const uint32_t win_shift = chunks.front().win_off;
    uint32_t last_end        = 0;
    for (DestuffChunk& c : chunks) {
        c.win_off -= win_shift;
        c.begin -= win_shift;
        c.end -= win_shift;
        c.dst_off -= dst_shift;
        if (c.pad_end) c.pad_end -= dst_shift;
        c.seg -= a;
        last_end = std::max(last_end, c.end);
    }
    s.xfer_begin += win_shift;
    s.xfer_end = s.xfer_begin + last_end;
    scan.segments.swap(segs);
    scan.chunks.swap(chunks);
    scan.num_subseq = scan.segments.back().subseq_offset + scan.segments.back().subseq_count;
    scan.tail_parts.clear();
    scan.tail_parts.push_back(0);
    for (const Segment& seg : scan.segments)
        if (seg.subseq_offset - scan.tail_parts.back() >= kTailPartSubseq) scan.tail_parts.push_back(seg.subseq_offset);
    scan.tail_parts.push_back(scan.num_subseq);
    return JPEGGPU_SUCCESS;
}

jpeggpu_status Reader::parse(const uint8_t* data, size_t size, int subseq_bytes, const Logger& log, bool device_scan,
                             int shard_rank, int shard_world)
{
    if (shard_world > 1 && device_scan) {
        // the share is cut out of the host walk's tables (jpeggpu_ext.h says so; the layout shows which walk was used)
        log.log("segment shard %d of %d: host walk, the device scan that was asked for is not used\n", shard_rank, shard_world);
        device_scan = false;
    }
    device_scan_ = device_scan;
    stop_        = false;
    {
        // reset, keeping the capacity of the per-scan vectors: a decoder parses image after image
        Scan keep[kMaxScans];
        for (int i = 0; i < kMaxScans; ++i) {
            keep[i].table_pack.swap(s.scans[i].table_pack);
            keep[i].table_pack_sync.swap(s.scans[i].table_pack_sync);
            keep[i].segments.swap(s.scans[i].segments);
            keep[i].chunks.swap(s.scans[i].chunks);
            keep[i].tail_parts.swap(s.scans[i].tail_parts);
        }
        s = Stream{};
        for (int i = 0; i < kMaxScans; ++i) {
            keep[i].table_pack.clear();
            keep[i].table_pack_sync.clear();
            keep[i].segments.clear();
            keep[i].chunks.clear();
            keep[i].tail_parts.clear();
            s.scans[i].table_pack.swap(keep[i].table_pack);
            s.scans[i].table_pack_sync.swap(keep[i].table_pack_sync);
            s.scans[i].segments.swap(keep[i].segments);
            s.scans[i].chunks.swap(keep[i].chunks);
            s.scans[i].tail_parts.swap(keep[i].tail_parts);
        }
    }
    std::memset(s.qtable, 0, sizeof(s.qtable));
    base_         = data;
    cur_          = data;
    end_          = data + size;
    subseq_request_ = subseq_bytes;
    subseq_bytes_   = subseq_bytes > 0 ? subseq_bytes : 64; // an automatic choice is made at the first scan header
    found_sof_    = false;
    std::memset(qt_defined_, 0, sizeof(qt_defined_));
    std::memset(dc_defined_, 0, sizeof(dc_defined_));
    std::memset(ac_defined_, 0, sizeof(ac_defined_));
    std::memset(comp_in_scan_, 0, sizeof(comp_in_scan_));
    if (size >= (1ull << 31)) return JPEGGPU_NOT_SUPPORTED;

    const auto read_marker = [&](uint8_t& marker) -> jpeggpu_status {
        if (remaining() < 2) {
            log.log("\ttoo few bytes for marker\n");
            return JPEGGPU_INVALID_JPEG;
        }
        const uint8_t ff = u8();
        if (ff != 0xFF) {
            log.log("\tinvalid marker byte 0x%02x\n", ff);
            return JPEGGPU_INVALID_JPEG;
        }
        marker = u8();
        while (marker == 0xFF) { // fill bytes
            if (remaining() < 1) return JPEGGPU_INVALID_JPEG;
            marker = u8();
        }
        return JPEGGPU_SUCCESS;
    };

    uint8_t marker = 0;
    jpeggpu_status st;
    if ((st = read_marker(marker)) != JPEGGPU_SUCCESS) return st;
    if (marker != M_SOI) return JPEGGPU_INVALID_JPEG;
    do {
        if ((st = read_marker(marker)) != JPEGGPU_SUCCESS) return st;
        log.log("marker 0x%02x\n", marker);
        st = JPEGGPU_SUCCESS;
        if (marker == M_SOF0 || marker == M_SOF1) {
            if (found_sof_) return JPEGGPU_INVALID_JPEG;
            found_sof_ = true;
            st         = read_sof(log);
        } else if (
            (marker >= 0xC2 && marker <= 0xCF && marker != M_DHT && marker != 0xC8 &&
             marker != 0xCC)) {
            log.log("\tunsupported JPEG type: SOF%d\n", marker - 0xC0);
            return JPEGGPU_NOT_SUPPORTED;
        } else if (marker == M_DHT) {
            st = read_dht(log);
        } else if (marker == M_SOS) {
            st = read_sos(log);
            if (st == JPEGGPU_SUCCESS && stop_) break;
        } else if (marker == M_DQT) {
            st = read_dqt(log);
        } else if (marker == M_DRI) {
            st = read_dri(log);
        } else if (marker == M_EOI) {
            break;
        } else if (marker == M_SOI || (marker >= M_RST0 && marker <= M_RST7) || marker == 0x01) {
            return JPEGGPU_INVALID_JPEG; // stand-alone markers have no place here
        } else {
            st = skip_segment(log);
        }
        if (st != JPEGGPU_SUCCESS) return st;
    } while (true);

    if (!found_sof_ || s.num_scans == 0) return JPEGGPU_INVALID_JPEG;
    for (int c = 0; c < s.num_comp; ++c) {
        if (!comp_in_scan_[c]) {
            log.log("\tcomponent with index %d not defined in scan\n", c);
            return JPEGGPU_INVALID_JPEG;
        }
    }
    if (shard_world > 1) return apply_segment_shard(shard_rank, shard_world, log);
    return JPEGGPU_SUCCESS;
}

} // namespace jg
