// emu_pipeline.cpp -- host emulation of the device pipeline's LOGIC, for CPU-only tests.
//
// Compiles the product's own parser (jg_reader.cpp) and per-lane symbol loop (jg_huff_core.h) with
// g++ and replays, lane by lane, what the gfx950 kernels of jg_kernels.hip do: destuff by work
// list, speculative decode, lock-step intra-sequence flows, inter-sequence flows, sequence tails,
// write pass with look-back. It has no LDS, no waves and no barriers -- each "iteration" loops over
// the lanes in order, reading the previous iteration's table -- so it checks the algorithm and the
// host-built tables, not the kernels' memory layout. The GPU tests check the kernels themselves.
#include "jg_bytes.h"
#include "jg_huff_core.h"
#include "jg_reader.hpp"

#include <cstring>
#include <vector>

using namespace jg;

namespace {

struct HostFetch {
    const uint8_t* seg; // first byte of the segment in the destuffed buffer
    int seg_words;
    typedef int Pos; // word of the segment
    Pos start(int w) const { return w; }
    void advance(Pos& q) const { ++q; }
    uint32_t load(const Pos& w) const
    {
        if (w < 0 || w >= seg_words) return 0; // the window starts one word early at bit 0 (BitWindow::seek)
        const uint8_t* p = seg + static_cast<size_t>(w) * 4;
        return static_cast<uint32_t>(p[0]) << 24 | p[1] << 16 | p[2] << 8 | p[3];
    }
    uint32_t cook(uint32_t v, const Pos&) const { return v; }
};

/// Host twin of the write pass's window (RowWindow in jg_kernels.hip): the segment's destuffed bytes, linear; the same
/// refill rule (not in two consecutive iterations), which the results must not depend on.
struct HostWindow {
    HostFetch f;
    uint32_t hi, lo, nxt;
    int pos, sh;
    bool ok;
    void seek(int p)
    {
        const int q = p - 1;
        sh          = 31 - (q & 31);
        pos         = q >> 5; // -1 for the word in front of the segment: loaded, never looked at
        hi          = f.load(pos);
        lo          = f.load(pos + 1);
        nxt         = f.load(pos + 2);
        pos += 2;
        ok = true;
    }
    void top()
    {
        const bool need = sh < 0 && ok;
        if (need) {
            hi  = lo;
            lo  = nxt;
            nxt = f.load(++pos);
            sh += 32;
        }
        ok = !need;
    }
    uint32_t look() const { return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (sh & 31)); }
    void skip(int n) { sh -= n; }
    int left() const { return sh; }
    int crossed() const { return 0; } // rows do not end here
    void cross() {}
    void done() {}
};

/// Host twin of the write pass's sink (StreamSink in jg_kernels.hip): symbol stream (jg_defs.h: 16-bit entries,
/// escapes behind coefficients that do not fit 10 bits) + data-unit table, fed by decode_units (jg_huff_core.h).
struct HostSink {
    uint16_t* sym;
    uint2_t* du_tab;
    uint32_t cur, cur_end, du_off;
    int du_index;
    int du, quota; // next data unit to start; first data unit of the next lane (or past the segment)
    bool started = false, open = false; // a DC symbol was seen; a unit's record is still to be written
    uint32_t unit_esc = 0;
    bool full() const { return du >= quota; }
    void unit_boundary()
    {
        if (open) du_tab[du_index] = uint2_t{du_off, (cur - du_off) | unit_esc};
        open = false;
    }
    void dc(int value)
    {
        started  = true;
        open     = true;
        du_off   = cur;
        du_index = du++;
        unit_esc = 0;
        if (cur < cur_end) sym[cur++] = static_cast<uint16_t>(static_cast<uint32_t>(value));
    }
    void ac(int category, int zpos, int value)
    {
        if (!started || category == 0) return; // the tail of the predecessor's unit; runs of zeros, ends of block, null entries
        if (cur < cur_end) sym[cur++] = static_cast<uint16_t>(sym_entry_ac(zpos, value));
    }
    void escape(int value)
    {
        if (!started) return;
        if (cur < cur_end) sym[cur++] = static_cast<uint16_t>(sym_entry_escape(value));
        unit_esc = kUnitHasEscape;
    }
    static constexpr int kFlushPeriod = 12;
    void flush_point(int) {}
};

struct St {
    int p, n, cz;
    uint32_t dc01, dc23;
};

void emu_destuff(const uint8_t* bytes, const Scan& sc, int subseq_bytes, std::vector<uint8_t>& dst, std::vector<int>& seg_idx)
{
    dst.assign(static_cast<size_t>(sc.num_subseq) * subseq_bytes + 256, 0xAA); // poison: padding must be written
    seg_idx.assign(sc.num_subseq, -1);
    for (const DestuffChunk& ck : sc.chunks) {
        uint32_t o = ck.dst_off;
        for (uint32_t pos = ck.begin; pos < ck.end; ++pos) {
            uint32_t p = pos > 0 ? bytes[pos - 1] : 0;
            if (ck.first && pos == ck.begin) p = 0;
            const uint32_t b = bytes[pos];
            if (p == 0xFF && b == 0) dst[o++] = 0xFF;
            else if (p != 0xFF && b != 0xFF) dst[o++] = static_cast<uint8_t>(b);
        }
        const uint32_t total = o - ck.dst_off;
        for (uint32_t z = o; ck.pad_end && z < ck.pad_end; ++z) dst[z] = 0;
        if (total) {
            for (uint32_t s = (ck.dst_off + subseq_bytes - 1) / subseq_bytes; s * subseq_bytes < ck.dst_off + total; ++s)
                seg_idx[s] = ck.seg;
        }
    }
}

} // namespace

static std::vector<int> g_write_iters, g_write_syms; // write pass, by subsequence: iterations of the lane's loop, symbols it decoded
static std::vector<St> g_pre_tail;       // studies (tools/probe/flow_study.py): the state table between the sequence
static std::vector<uint8_t> g_pre_pend;  // kernel and the tail pass, and its pending marks

extern "C" {

/// The state table as the tail pass found it in the last emu_decode_scan (p, c | z << 8, pending mark); returns its length.
int emu_read_pre_tail(int* p, int* cz, uint8_t* pend, int cap)
{
    const int n = static_cast<int>(g_pre_tail.size());
    for (int i = 0; i < n && i < cap; ++i) {
        p[i]    = g_pre_tail[i].p;
        cz[i]   = g_pre_tail[i].cz;
        pend[i] = g_pre_pend[i];
    }
    return n;
}

/// The product's symbol step on 32-bit windows, for the reference-built known answers (tests/test_huff_kats.py): the
/// table is built from a DHT payload by the parser's own build_huff_table (and widened for the sync pack), looked up the
/// way the kernels do (jg_huff_core.h: lut16_entry, huff_second_level, huff_long_code, bits_field, extend_bits), and the
/// result is expressed as the reference's decode_next_symbol reports it. `path`: 0 write pack, 1 sync pack (low halves
/// of the 32-bit entries), 2 the long-code path for every window (whatever the first level says).
int emu_symbol_steps(const uint8_t* bits, const uint8_t* vals, int count, int is_dc, int path, const uint32_t* win, const int* z, int n,
                     int* length, int* symbol, int* run)
{
    uint8_t nc[16];
    std::memcpy(nc, bits, 16);
    std::vector<uint8_t> t, wide;
    build_huff_table(t, nc, vals, count, is_dc != 0);
    widen_huff_table(t, is_dc != 0, wide);
    const bool dc = is_dc != 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t peek = win[i];
        uint32_t e;
        if (path == 0) {
            e = dc ? lut16_entry<kLutBitsDc>(t.data(), peek) : lut16_entry<kLutBitsAc>(t.data(), peek);
            if ((e & 31u) == 0) e = huff_second_level(t.data(), e, peek, dc);
        } else if (path == 1) {
            const uint32_t idx = peek >> (dc ? 32 - kLutBitsDc : 32 - kLutBitsAc);
            e                  = ld_u16(wide.data() + kSyncEntryBytes * idx);
            if ((e & 31u) == 0) e = huff_second_level<kSyncEntryBytes>(wide.data(), e, peek, dc);
        } else {
            e = huff_long_code(t.data() + (dc ? (2 << kLutBitsDc) : (2 << kLutBitsAc)), peek, dc);
        }
        const int total = e & 31, s = (e >> 5) & 15, adv = static_cast<int>(e >> 9);
        length[i] = total;
        symbol[i] = s ? extend_bits(bits_field(peek, total, s), s) : 0;
        run[i]    = dc ? 0 : s ? adv - 1 : adv == static_cast<int>(kEobAdvance) ? 63 - z[i] : adv - 1; // an end of block's advance (jg_defs.h)
    }
    return 0;
}

/// The product's byte rule (jg_bytes.h, as destuff_kernel applies it to four bytes at a time) on (previous byte, byte)
/// pairs, with the pair placed inside one word (`across` 0) or across two words (1: the carry between words).
void emu_byte_rule(const uint8_t* prev, const uint8_t* byte, int n, int across, uint8_t* is_data, uint8_t* written)
{
    for (int i = 0; i < n; ++i) {
        // bytes in front of and behind the pair are plain data (0x11) so that only the pair decides
        uint32_t w0 = across ? 0x11111111u : 0x11111111u, w1 = 0x11111111u;
        int at; // byte index of `byte` in w1
        if (across) {
            w0 = (w0 & 0x00FFFFFFu) | static_cast<uint32_t>(prev[i]) << 24;
            w1 = (w1 & 0xFFFFFF00u) | byte[i];
            at = 0;
        } else {
            w1 = (w1 & 0xFF0000FFu) | static_cast<uint32_t>(prev[i]) << 8 | static_cast<uint32_t>(byte[i]) << 16;
            at = 2;
        }
        const uint32_t F0 = bytes_ff(w0), F1 = bytes_ff(w1), Z1 = bytes_zero(w1);
        const uint32_t PF      = of_previous_byte(F1, F0);
        const uint32_t stuffed = PF & Z1;
        const uint32_t data    = stuffed | (~(PF | F1) & kHi80);
        w1 |= spread80(stuffed);
        is_data[i] = (collapse80(data) >> at) & 1u;
        written[i] = is_data[i] ? static_cast<uint8_t>(w1 >> (8 * at)) : 0;
    }
}

/// Iterations and symbols of every lane of the write pass of the last emu_decode_scan (tools/probe/write_study.py).
int emu_read_write_iters(int* iters, int* syms, int cap)
{
    const int n = static_cast<int>(g_write_iters.size());
    for (int i = 0; i < n && i < cap; ++i) {
        iters[i] = g_write_iters[i];
        syms[i]  = g_write_syms[i];
    }
    return n;
}

int g_active_hist[512];
int g_multi_hypothesis = 0; // emu_set_multi_hypothesis: the lone-decode path with the multi-hypothesis table (jg_defs.h)
int g_mh_known = 0, g_mh_subseq = 0; // of the last scan decoded that way: entries the chain supplied / subsequences
void emu_set_multi_hypothesis(int on) { g_multi_hypothesis = on; }

/// Returns a jpeggpu_status. Outputs are for scan `scan_idx`; pointers may be null.
int emu_decode_scan(
    const uint8_t* data,
    size_t size,
    int subseq_bytes,
    int max_intra_iters, // cap of the lock-step loop of huff_sync_intra (kSeqLanes = no cap)
    int scan_idx,
    int* out_num_subseq,
    int* out_num_du,
    uint8_t* destuffed, // [num_subseq * subseq_bytes]
    int* seg_index,     // [num_subseq]
    int* st_p,
    int* st_n,
    int* st_cz,
    int* st_dc, // [4][num_subseq]
    int16_t* coef,
    int* out_max_flow_iters)
{
    Reader rd;
    Logger log;
    const jpeggpu_status stat = rd.parse(data, size, subseq_bytes, log);
    if (stat != JPEGGPU_SUCCESS) return stat;
    const Stream& s = rd.s;
    if (scan_idx < 0 || scan_idx >= s.num_scans) return JPEGGPU_INVALID_ARGUMENT;
    const Scan& sc = s.scans[scan_idx];
    if (out_num_subseq) *out_num_subseq = sc.num_subseq;
    if (out_num_du) *out_num_du = sc.num_du;
    if (!coef) return JPEGGPU_SUCCESS;

    // the transferred byte buffer, as the device sees it
    std::vector<uint8_t> bytes(s.xfer_end - s.xfer_begin + 2 * kDestuffWin, 0);
    std::memcpy(bytes.data(), data + s.xfer_begin, s.xfer_end - s.xfer_begin);

    std::vector<uint8_t> dst;
    std::vector<int> segi;
    emu_destuff(bytes.data(), sc, subseq_bytes, dst, segi);

    ScanParams sp{};
    sp.num_subseq       = sc.num_subseq;
    sp.num_segments     = static_cast<int>(sc.segments.size());
    sp.du_per_mcu       = sc.du_per_mcu;
    sp.num_comp         = sc.num_comp;
    sp.mcus_per_segment = sc.mcus_per_segment;
    sp.total_mcus       = sc.mcus_x * sc.mcus_y;
    sp.subseq_words     = subseq_bytes / 4;
    sp.tab_bytes        = static_cast<uint32_t>(sc.table_pack.size());
    sp.cursor_off       = sc.cursor_off;
    sp.tab_bytes_sync   = static_cast<uint32_t>(sc.table_pack_sync.size());
    sp.cursor_off_sync  = sc.cursor_off_sync;
    const uint8_t* tabs = sc.table_pack.data();           // write pass
    const uint8_t* tabs_sync = sc.table_pack_sync.data(); // state-only passes (multi-symbol entries)
    ScanParams sp_sync  = sp;
    sp_sync.use_sync_pack();

    const int S    = sc.num_subseq;
    const int T    = max_intra_iters >= kSeqLanes ? kSeqSubseq : kSeqSubseqBatch; // a lone decode's sequences, or a batch's (jg_defs.h)
    const int bits = subseq_bytes * 8;
    const int W    = subseq_bytes / 4;
    std::vector<St> st(S);
    NoSink nosink;
    int max_iters = 0;

    struct Lane {
        LaneState s;
        BitWindow<HostFetch> bw;
        HostFetch f;
        int end_bit, lim;
        bool flowing;
    };

    std::vector<uint8_t> pend(S, 0);
    if (max_intra_iters < 1) return JPEGGPU_INVALID_ARGUMENT; // the first flow iteration supplies n and the DC sums

    // ---- speculative pass of every subsequence (first half of huff_sync_intra): exit state only ----
    for (int sub = 0; sub < S; ++sub) {
        const Segment seg = sc.segments[segi[sub]];
        const int rel     = sub - seg.subseq_offset;
        HostFetch f{dst.data() + static_cast<size_t>(seg.subseq_offset) * subseq_bytes, seg.subseq_count * W};
        LaneState ls{};
        ls.p = rel * bits;
        BitWindow<HostFetch> bw;
        bw.seek(ls.p, f);
        SpecSink spec_sink;
        decode_subsequence(ls, bw, f, (rel + 1) * bits, tabs_sync, sp_sync, spec_sink);
        st[sub].p  = ls.p;
        st[sub].cz = ls.c | (ls.z << 8);
    }
    // ---- multi-hypothesis speculation (huff_mh_spec / _flow / _resolve): the table the flows below start from ----
    std::vector<uint8_t> known(S, 1);
    bool mh = false;
    g_mh_known = g_mh_subseq = 0;
    if (g_multi_hypothesis && max_intra_iters >= kSeqLanes && sc.du_per_mcu >= 2 && sc.du_per_mcu <= kMhMaxHyp) {
        // (segments of more than kMhMaxSegSubseq subsequences are walked block-wise on the device -- up to kMhMaxBlocks
        // blocks --, which is the same chain: here it is followed link by link whatever its length)
        int longest = 0, blocks = 0;
        for (const Segment& g : sc.segments) {
            longest = std::max(longest, g.subseq_count);
            blocks += (g.subseq_count + kMhMaxSegSubseq - 1) / kMhMaxSegSubseq;
        }
        mh = longest <= kMhMaxSegSubseq || blocks <= kMhMaxBlocks;
    }
    if (mh) {
        const int H = sc.du_per_mcu;
        const auto decode_from = [&](int sub, int p, int cz) { // exit state of `sub` decoded from (p, c, z)
            const Segment seg = sc.segments[segi[sub]];
            const int rel     = sub - seg.subseq_offset;
            HostFetch f{dst.data() + static_cast<size_t>(seg.subseq_offset) * subseq_bytes, seg.subseq_count * W};
            LaneState ls{};
            ls.p = p;
            ls.c = cz & 0xFF;
            ls.z = cz >> 8;
            BitWindow<HostFetch> bw;
            bw.seek(ls.p, f);
            SpecSink spec_sink;
            decode_subsequence(ls, bw, f, (rel + 1) * bits, tabs_sync, sp_sync, spec_sink);
            return St{ls.p, 0, ls.c | (ls.z << 8), 0u, 0u};
        };
        std::vector<St> cand(static_cast<size_t>(H) * S);
        for (int sub = 0; sub < S; ++sub)
            for (int h = 0; h < H; ++h) cand[static_cast<size_t>(h) * S + sub] = decode_from(sub, (sub - sc.segments[segi[sub]].subseq_offset) * bits, h);
        struct Link { int k, g; std::vector<St> passed; };
        const auto link_of = [&](int sub, int h) {
            Link l{0, 0, {}};
            const Segment seg = sc.segments[segi[sub]];
            St x = cand[static_cast<size_t>(h) * S + sub];
            for (int k = 1; k <= kMhSteps; ++k) {
                const int t = sub + k;
                if (t >= S) break;
                if (t >= seg.subseq_offset + seg.subseq_count) { l.k = k; l.g = 0; break; }
                x = decode_from(t, x.p, x.cz);
                int g = -1;
                for (int q = 0; q < H; ++q)
                    if (cand[static_cast<size_t>(q) * S + t].p == x.p && cand[static_cast<size_t>(q) * S + t].cz == x.cz) g = q;
                if (g >= 0) { l.k = k; l.g = g; break; }
                l.passed.push_back(x);
            }
            if (l.k == 0) l.passed.clear();
            return l;
        };
        g_mh_known = 0;
        g_mh_subseq = S;
        for (const Segment& seg : sc.segments) {
            const int n = seg.subseq_count, base = seg.subseq_offset;
            for (int r = 0; r < n; ++r) { st[base + r] = cand[base + r]; known[base + r] = 0; } // placeholders: hypothesis 0
            int r = 0, h = 0, broke = n;
            while (r < n) {
                st[base + r]    = cand[static_cast<size_t>(h) * S + base + r];
                known[base + r] = 1;
                const Link l    = link_of(base + r, h);
                if (l.k == 0) { broke = r + 1; break; }
                for (int q = 1; q < l.k && r + q < n; ++q) { st[base + r + q] = l.passed[q - 1]; known[base + r + q] = 1; }
                r += l.k;
                h = l.g;
            }
            for (int q = broke; q < n; ++q) known[base + q] = 1; // plain speculation behind a break
            for (int q = 0; q < n; ++q) g_mh_known += known[base + q] && q < broke;
        }
    }
    const std::vector<St> spec = st;

    // ---- flow passes inside a sequence (second half of huff_sync_intra) ----
    // Emulation lane t first decodes subsequence first + t (the device's lane index is one lower: there a
    // lane holds the predecessor): from the predecessor's speculative exit state, or from the segment's
    // start state when the subsequence opens a segment.
    const int num_seq = (S + T - 1) / T;
    for (int b = 0; b < num_seq; ++b) {
        const int first = b * T, nsub = std::min(T, S - first);
        std::vector<Lane> ln(nsub);
        for (int t = 0; t < nsub; ++t) {
            const int j        = first + t;
            const Segment seg  = sc.segments[segi[j]];
            const int rel      = j - seg.subseq_offset;
            Lane& L            = ln[t];
            L.f                = HostFetch{dst.data() + static_cast<size_t>(seg.subseq_offset) * subseq_bytes, seg.subseq_count * W};
            L.lim              = std::min(nsub, seg.subseq_offset + seg.subseq_count - first);
            L.s                = LaneState{};
            if (rel > 0) {
                L.s.p = spec[j - 1].p;
                L.s.c = spec[j - 1].cz & 0xFF;
                L.s.z = spec[j - 1].cz >> 8;
            }
            L.end_bit = rel * bits;
            L.bw.seek(L.s.p, L.f);
            L.flowing = rel == 0 || known[j - 1] != 0; // an entry the chain hopped over starts no flow
        }
        int iter = 0;
        for (; iter < max_intra_iters; ++iter) {
            bool any = false;
            for (int t = 0; t < nsub; ++t) if (ln[t].flowing && t + iter < ln[t].lim && iter < 512) ++g_active_hist[iter];
            for (int t = 0; t < nsub; ++t) { // entry t + iter of iteration `iter` is touched by lane t only
                Lane& L     = ln[t];
                const int j = t + iter;
                if (L.flowing && j < L.lim) {
                    L.s.n = 0;
                    L.s.dc01 = L.s.dc23 = 0;
                    L.end_bit += bits;
                    decode_subsequence(L.s, L.bw, L.f, L.end_bit, tabs_sync, sp_sync, nosink);
                    St& o        = st[first + j];
                    const int cz = L.s.c | (L.s.z << 8);
                    if (L.s.p == o.p && cz == o.cz && known[first + j]) L.flowing = false; // ... and stops none
                    o.p = L.s.p; o.n = L.s.n; o.cz = cz;
                    o.dc01 = L.s.dc01; o.dc23 = L.s.dc23;
                    known[first + j] = 1;
                } else {
                    L.flowing = false;
                }
                any |= L.flowing && j + 1 < L.lim;
            }
            if (iter + 1 > max_iters) max_iters = iter + 1;
            if (!any) break;
        }
        if (iter == max_intra_iters) { // flows cut short continue in the tail pass from the entry they reached
            for (int t = 0; t < nsub; ++t)
                if (ln[t].flowing && t + iter < ln[t].lim) pend[first + t + iter - 1] = 1;
        }
    }

    g_pre_tail.assign(st.begin(), st.end());
    g_pre_pend = pend;
    // ---- tail pass (huff_sync_tail): sequence boundaries + pending flows, per part, groups of 256 ----
    for (size_t part = 0; part + 1 < sc.tail_parts.size(); ++part) {
        const int lo = sc.tail_parts[part], hi = sc.tail_parts[part + 1];
        std::vector<int> list;
        for (int sub = lo; sub < hi; ++sub) {
            if (sub + 1 >= S) continue;
            // a sequence boundary needs a flow only where the state the next sequence's first entry was decoded from (here:
            // the speculative table; the device's workgroups record theirs, bnd_p / bnd_cz) is not the stored one
            const bool boundary = (sub + 1) % T == 0 && segi[sub] == segi[sub + 1] && (spec[sub].p != st[sub].p || spec[sub].cz != st[sub].cz);
            if (pend[sub] || boundary) list.push_back(sub);
        }
        for (size_t g = 0; g < list.size(); g += 256) {
            const int nb = static_cast<int>(std::min<size_t>(256, list.size() - g));
            std::vector<Lane> ln(nb);
            std::vector<int> jj(nb);
            for (int t = 0; t < nb; ++t) {
                const int from    = list[g + t];
                const Segment seg = sc.segments[segi[from]];
                Lane& L           = ln[t];
                L.lim             = seg.subseq_offset + seg.subseq_count;
                jj[t]             = from + 1;
                L.flowing         = jj[t] < L.lim;
                if (L.flowing) {
                    L.f       = HostFetch{dst.data() + static_cast<size_t>(seg.subseq_offset) * subseq_bytes, seg.subseq_count * W};
                    L.s       = LaneState{};
                    L.s.p     = st[from].p;
                    L.s.c     = st[from].cz & 0xFF;
                    L.s.z     = st[from].cz >> 8;
                    L.end_bit = (from - seg.subseq_offset + 1) * bits;
                    L.bw.seek(L.s.p, L.f);
                }
            }
            while (true) {
                bool any = false;
                for (int t = 0; t < nb; ++t) {
                    Lane& L = ln[t];
                    int& j  = jj[t];
                    if (L.flowing && j < L.lim) {
                        L.s.n = 0;
                        L.s.dc01 = L.s.dc23 = 0;
                        L.end_bit += bits;
                        decode_subsequence(L.s, L.bw, L.f, L.end_bit, tabs_sync, sp_sync, nosink);
                        St& o        = st[j];
                        const int cz = L.s.c | (L.s.z << 8);
                        if (L.s.p == o.p && cz == o.cz) L.flowing = false;
                        o.p = L.s.p; o.n = L.s.n; o.cz = cz;
                        o.dc01 = L.s.dc01; o.dc23 = L.s.dc23;
                        ++j;
                    } else {
                        L.flowing = false;
                    }
                    any |= L.flowing && j < L.lim;
                }
                if (!any) break;
            }
        }
    }

    // ---- sequence tails (huff_seq_tails) ----
    std::vector<St> tails(num_seq);
    for (int b = 0; b < num_seq; ++b) {
        const int first = b * T, nsub = std::min(T, S - first);
        const int open_from = sc.segments[segi[first + nsub - 1]].subseq_offset;
        St acc{};
        for (int t = 0; t < nsub; ++t) {
            if (first + t >= open_from) {
                acc.n += st[first + t].n;
                acc.dc01 = pk_add_u16(acc.dc01, st[first + t].dc01);
                acc.dc23 = pk_add_u16(acc.dc23, st[first + t].dc23);
            }
        }
        tails[b] = acc;
    }

    g_write_iters.assign(S, 0);
    g_write_syms.assign(S, 0);
    // ---- write pass (huff_write): symbol stream + data-unit table, then gather like the IDCT does ----
    const uint32_t region = sym_region_entries(subseq_bytes);
    std::vector<uint16_t> sym(static_cast<size_t>(S) * region, 0xDEADu);
    std::vector<uint2_t> du_tab(sc.num_du, uint2_t{0xFFFFFFFFu, 0xFFFFFFFFu});
    for (int b = 0; b < num_seq; ++b) {
        const int first = b * T, nsub = std::min(T, S - first);
        const Segment seg0 = sc.segments[segi[first]];
        St carry{};
        if (seg0.subseq_offset < first) {
            for (int a = seg0.subseq_offset / T; a < b; ++a) {
                carry.n += tails[a].n;
                carry.dc01 = pk_add_u16(carry.dc01, tails[a].dc01);
                carry.dc23 = pk_add_u16(carry.dc23, tails[a].dc23);
            }
        }
        std::vector<St> ex(nsub + 1);
        St run{};
        for (int t = 0; t < nsub; ++t) {
            ex[t] = run;
            run.n += st[first + t].n;
            run.dc01 = pk_add_u16(run.dc01, st[first + t].dc01);
            run.dc23 = pk_add_u16(run.dc23, st[first + t].dc23);
        }
        for (int t = 0; t < nsub; ++t) {
            const int sub      = first + t;
            const int seg_i    = segi[sub];
            const Segment seg  = sc.segments[seg_i];
            const int rel      = sub - seg.subseq_offset;
            const bool carried = seg.subseq_offset < first;
            const int ts       = carried ? 0 : seg.subseq_offset - first;
            HostSink sink;
            sink.sym     = sym.data();
            sink.du_tab  = du_tab.data();
            sink.cur     = static_cast<uint32_t>(sub) * region;
            sink.cur_end = sink.cur + region;
            sink.du_off  = sink.cur;
            sink.du_index = 0;
            const int nprefix = ex[t].n - ex[ts].n + (carried ? carry.n : 0);
            const auto sub16 = [](uint32_t a, uint32_t b) { return pk_add_u16(a, pk_add_u16(~b, 0x00010001u)); };
            const int m0 = seg_i * sp.mcus_per_segment, m1 = std::min(m0 + sp.mcus_per_segment, sp.total_mcus);
            sink.du    = m0 * sp.du_per_mcu + ((nprefix + 63) >> 6);
            // the lane stops in front of the next lane's first unit, or at the segment's quota if it is the last
            sink.quota = m1 * sp.du_per_mcu;
            if (rel + 1 < seg.subseq_count) sink.quota = std::min(sink.quota, m0 * sp.du_per_mcu + ((nprefix + st[sub].n + 63) >> 6));
            LaneState ls{};
            ls.dc01 = pk_add_u16(sub16(ex[t].dc01, ex[ts].dc01), carried ? carry.dc01 : 0u); // predictors so far
            ls.dc23 = pk_add_u16(sub16(ex[t].dc23, ex[ts].dc23), carried ? carry.dc23 : 0u);
            if (rel > 0) {
                ls.p = st[sub - 1].p;
                ls.c = st[sub - 1].cz & 0xFF;
                ls.z = st[sub - 1].cz >> 8;
            }
            HostWindow win{HostFetch{dst.data() + static_cast<size_t>(seg.subseq_offset) * subseq_bytes, seg.subseq_count * W}};
            int it[4] = {0, 0, 0, 0};
            decode_units(ls, win, tabs, sp, sink, 2 * (bits + 64 * 32), it);
            g_write_iters[sub] = it[0];
            g_write_syms[sub]  = it[1];
            sink.unit_boundary(); // nothing is left open when the lane stops (it stops in a DC slot), but say so
        }
    }
    {
        static const uint8_t nat[64] = JG_ORDER_NATURAL;
        std::memset(coef, 0, static_cast<size_t>(sc.num_du) * 128);
        for (int d = 0; d < sc.num_du; ++d) {
            const uint2_t e = du_tab[d];
            const uint32_t n = e.y & 0x7Fu;
            if (e.y > 0xFFu || n < 1 || static_cast<size_t>(e.x) + n > sym.size()) return JPEGGPU_INTERNAL_ERROR; // table entry never written
            coef[static_cast<size_t>(d) * 64] = static_cast<int16_t>(sym[e.x]); // the first entry is the DC value
            bool escaped = false;
            for (uint32_t k = 1; k < n; ++k) {
                const uint32_t v = sym[e.x + k];
                if (sym_entry_index(v) == 0) continue; // an escape: taken with the entry in front of it
                const uint32_t nx = k + 1 < n ? sym[e.x + k + 1] : 1u;
                const bool esc    = sym_entry_index(nx) == 0;
                escaped |= esc;
                coef[static_cast<size_t>(d) * 64 + nat[sym_entry_index(v)]] = static_cast<int16_t>(esc ? sym_entry_value(v, nx) : sym_entry_value(v));
            }
            if (escaped != ((e.y & kUnitHasEscape) != 0)) return JPEGGPU_INTERNAL_ERROR; // the record's flag says what the entries hold
        }
    }

    if (destuffed) std::memcpy(destuffed, dst.data(), static_cast<size_t>(S) * subseq_bytes);
    if (seg_index) std::memcpy(seg_index, segi.data(), sizeof(int) * S);
    for (int i = 0; i < S; ++i) {
        if (st_p) st_p[i] = st[i].p;
        if (st_n) st_n[i] = st[i].n;
        if (st_cz) st_cz[i] = st[i].cz;
        if (st_dc) { // unpacked (sign-extended 16-bit sums) for comparison with the oracle modulo 2^16
            st_dc[0 * static_cast<size_t>(S) + i] = static_cast<int16_t>(st[i].dc01 & 0xFFFF);
            st_dc[1 * static_cast<size_t>(S) + i] = static_cast<int16_t>(st[i].dc01 >> 16);
            st_dc[2 * static_cast<size_t>(S) + i] = static_cast<int16_t>(st[i].dc23 & 0xFFFF);
            st_dc[3 * static_cast<size_t>(S) + i] = static_cast<int16_t>(st[i].dc23 >> 16);
        }
    }
    if (out_max_flow_iters) *out_max_flow_iters = max_iters;
    return JPEGGPU_SUCCESS;
}
}
