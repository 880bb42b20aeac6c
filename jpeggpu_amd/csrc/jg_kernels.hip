// jg_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the baseline JPEG decode path.
//
//   destuff_kernel          byte-stuffing / restart-marker removal   (reference src/decode_destuff.cu:37-361)
//   huff_sync_intra         speculative decode + intra-sequence sync (reference decode_huffman.cu:413-524)
//   huff_sync_inter         inter-sequence sync                      (reference decode_huffman.cu:534-621)
//   huff_seq_tails          per-sequence sums of n / DC              (replaces cub ExclusiveScanByKey :818-869
//                                                                     and the DC scans of decode_dc.cu:88-169)
//   huff_write              final decode, de-zigzag, absolute DC     (reference decode_huffman.cu:627-682)
//   idct_kernel             dequant + 8x8 fixed-point IDCT reading stream order
//                                                                    (reference idct.cu:44-223 + decode_transpose.cu:41-132)
//
// Everything is integer / bit-serial: no MFMA. The bitstream slice of a workgroup is staged through
// LDS with coalesced 4-byte global loads into a padded [word][subsequence] layout so that the 64
// lanes of a wave, each walking its own subsequence, hit 64 different banks.
#include "jg_huff_core.h"
#include "jg_kernels.hpp"

#include <hip/hip_runtime.h>

namespace jg {

namespace {

constexpr int T = kSeqSubseq; // lanes (= subsequences) per workgroup in the Huffman kernels

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if (lane_id() >= d) v += o;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// destuff
// ------------------------------------------------------------------------------------------------

/// One workgroup = one aligned 4 KiB window of the transferred bytes clipped to one segment.
/// Lane t owns 16 consecutive source bytes. Byte rule (reference src/decode_destuff.cu:37-44): a byte is
/// data iff (prev == FF and b == 00) or (prev != FF and b != FF); the first case stores FF.
/// The compacted bytes are staged in LDS at the destination's 16-byte phase and leave as whole
/// 16-byte stores except at the two ragged ends (neighbouring chunks own the other bytes there).
__global__ __launch_bounds__(256) void destuff_kernel(
    const uint8_t* __restrict__ src,
    uint8_t* __restrict__ dst,
    int* __restrict__ seg_idx,
    const DestuffChunk* __restrict__ chunks,
    int subseq_shift)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_out[kDestuffWin + 32];
    __shared__ uint32_t s_wave[4];

    const DestuffChunk ck = chunks[blockIdx.x];
    const int t           = threadIdx.x;
    const uint32_t gpos   = ck.win_off + t * 16;

    uint32_t w[4];
    {
        const uint4 v = *reinterpret_cast<const uint4*>(src + gpos);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    }
    uint32_t prev = __shfl_up(w[3] >> 24, 1);
    if (lane_id() == 0) prev = gpos > 0 ? src[gpos - 1] : 0u;

    uint32_t mask = 0; // bit i: byte i is data
    uint32_t ffm  = 0; // bit i: byte i is stored as FF
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t b   = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const uint32_t pos = gpos + i;
        uint32_t p         = prev;
        if (ck.first && pos == ck.begin) p = 0; // predecessor is a marker byte, never stuffing
        const bool in      = pos >= ck.begin && pos < ck.end;
        const bool stuffed = p == 0xFFu && b == 0u;
        const bool plain   = p != 0xFFu && b != 0xFFu;
        if (in && (stuffed || plain)) mask |= 1u << i;
        if (stuffed) ffm |= 1u << i;
        prev = b;
    }

    const uint32_t cnt  = __popc(mask);
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane_id() == 63) s_wave[t >> 6] = incl;
    __syncthreads();
    uint32_t off = incl - cnt;
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t ws = s_wave[k];
        if (k < (t >> 6)) off += ws;
        total += ws;
    }

    const uint32_t phase = ck.dst_off & 15u;
    {
        uint32_t o = phase + off;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (mask & (1u << i)) {
                const uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                s_out[o++]       = static_cast<uint8_t>((ffm >> i) & 1u ? 0xFFu : b);
            }
        }
    }
    __syncthreads();

    // write-out: 16-byte granules of the destination, aligned
    uint8_t* const dbase = dst + (ck.dst_off - phase);
    const uint32_t lo = phase, hi = phase + total; // valid LDS byte range
    for (uint32_t g = t; g * 16 < hi; g += 256) {
        const uint32_t b0 = g * 16;
        if (b0 >= lo && b0 + 16 <= hi) {
            *reinterpret_cast<uint4*>(dbase + b0) = *reinterpret_cast<const uint4*>(s_out + b0);
        } else {
            for (uint32_t b = b0 > lo ? b0 : lo; b < b0 + 16 && b < hi; ++b) dbase[b] = s_out[b];
        }
    }
    // zero the tail of the segment up to its subsequence-aligned end
    if (ck.pad_end) {
        for (uint32_t b = ck.dst_off + total + t; b < ck.pad_end; b += 256) dst[b] = 0;
    }
    // subsequences that start inside this chunk's destination range belong to this segment
    if (total) {
        const uint32_t sb    = 1u << subseq_shift;
        const uint32_t first = (ck.dst_off + sb - 1) >> subseq_shift;
        for (uint32_t s = first + t; (s << subseq_shift) < ck.dst_off + total; s += 256) seg_idx[s] = ck.seg;
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: bitstream access
// ------------------------------------------------------------------------------------------------

/// LDS image of one sequence's bitstream: word k of local subsequence t lives at
/// k * (T + PAD) + t, PAD = 32 / W (W = words per subsequence, at most 32 here) so that both the
/// coalesced fill (consecutive k) and the decode-time reads (consecutive t) are bank-conflict-free.
template <int W>
struct SeqImage {
    static constexpr int kPad    = W >= 32 ? 1 : 32 / W;
    static constexpr int kStride = T + kPad;
    static constexpr int kWords  = W * kStride;
};

/// `edge` holds the three words around the sequence: [0] the word before it (the write pass of the
/// sequence's first subsequence starts up to 31 bits before its own first bit, reference
/// decode_huffman_reader.hpp:279-292 carries those bits in `cache`), [1] and [2] the two words after
/// it (a symbol may be peeked across the sequence's end).
template <int W>
struct LdsFetch {
    const uint32_t* img;
    const uint32_t* edge;
    int base;      // word offset of the segment's first word relative to the sequence's first word
    int seg_words; // words in the segment (zero beyond, reference decode_huffman_reader.hpp:110-152)
    __device__ __forceinline__ uint32_t operator()(int w) const
    {
        if (w >= seg_words) return 0u;
        const int local = base + w;
        if (local < 0) return edge[0];
        const int t = local / W; // W is a power of two
        const int k = local % W;
        if (t < T) return img[k * SeqImage<W>::kStride + t];
        return edge[1 + ((local - T * W) & 1)];
    }
};

struct GlobalFetch {
    const uint32_t* words; // first word of the segment
    int seg_words;
    __device__ __forceinline__ uint32_t operator()(int w) const
    {
        return w < seg_words ? __builtin_bswap32(words[w]) : 0u;
    }
};

template <int W>
__device__ __forceinline__ void load_sequence(
    uint32_t* img, uint32_t* edge, const uint32_t* __restrict__ scan32, int first_sub, int nsub, int num_subseq)
{
    const uint32_t* src = scan32 + static_cast<size_t>(first_sub) * W;
    const int nwords    = nsub * W;
    for (int i = threadIdx.x; i < nwords; i += T) {
        img[(i % W) * SeqImage<W>::kStride + i / W] = __builtin_bswap32(src[i]);
    }
    if (threadIdx.x < 2) {
        const bool more       = first_sub + nsub < num_subseq;
        edge[1 + threadIdx.x] = more ? __builtin_bswap32(src[nwords + threadIdx.x]) : 0u;
    }
    if (threadIdx.x == 2) edge[0] = first_sub > 0 ? __builtin_bswap32(src[-1]) : 0u;
}

__device__ __forceinline__ void load_tables(HuffTableDev* s_tab, const HuffTableDev* __restrict__ g_tab)
{
    constexpr int n    = kHuffSlots * sizeof(HuffTableDev) / 4;
    uint32_t* d        = reinterpret_cast<uint32_t*>(s_tab);
    const uint32_t* s  = reinterpret_cast<const uint32_t*>(g_tab);
    for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
}

// ------------------------------------------------------------------------------------------------
// Huffman: speculative decode + intra-sequence synchronisation
// ------------------------------------------------------------------------------------------------

/// Lane t decodes subsequence t of the sequence from the guessed state (c, z) = (0, 0), then keeps
/// flowing into subsequences t+1, t+2, ... of the same segment until the state it reaches equals the
/// one stored there. In iteration i entry j = t+1+i of the LDS state table is
/// read and written by lane t only, so one workgroup barrier per iteration is enough.
template <int W>
__global__ __launch_bounds__(T) void huff_sync_intra(
    const uint32_t* __restrict__ scan32,
    const Segment* __restrict__ segments,
    const int* __restrict__ seg_idx,
    const HuffTableDev* __restrict__ g_tables,
    ScanParams sp,
    SubseqState out)
{
    __shared__ uint32_t s_img[SeqImage<W>::kWords];
    __shared__ uint32_t s_tail[3];
    __shared__ HuffTableDev s_tab[kHuffSlots];
    __shared__ int s_p[T], s_n[T], s_cz[T], s_dc[kMaxComp][T];

    const int t         = threadIdx.x;
    const int first_sub = blockIdx.x * T;
    const int nsub      = min(T, sp.num_subseq - first_sub);

    load_tables(s_tab, g_tables);
    load_sequence<W>(s_img, s_tail, scan32, first_sub, nsub, sp.num_subseq);
    __syncthreads();

    const bool active = t < nsub;
    LaneState st{};
    BitWindow<LdsFetch<W>> bw{};
    LdsFetch<W> fetch{s_img, s_tail, 0, 0};
    int end_bit = 0;
    int lim     = 0; // flows stay below this local index: end of the segment or of the sequence
    NoSink sink;
    if (active) {
        const int sub     = first_sub + t;
        const Segment seg = segments[seg_idx[sub]];
        const int rel     = sub - seg.subseq_offset;
        fetch.base        = (seg.subseq_offset - first_sub) * W;
        fetch.seg_words   = seg.subseq_count * W;
        lim               = min(nsub, seg.subseq_offset + seg.subseq_count - first_sub);
        st.p              = rel * (W * 32);
        end_bit           = (rel + 1) * (W * 32);
        bw.seek(st.p, fetch);
        decode_subsequence(st, bw, fetch, end_bit, s_tab, sp, sink);
        s_p[t]  = st.p;
        s_n[t]  = st.n;
        s_cz[t] = st.c | (st.z << 8);
#pragma unroll
        for (int k = 0; k < kMaxComp; ++k) s_dc[k][t] = st.dc[k];
    }
    __syncthreads();

    bool flowing = active;
    for (int iter = 0; iter < T; ++iter) {
        const int j = t + 1 + iter;
        if (flowing && j < lim) {
            st.n = 0;
#pragma unroll
            for (int k = 0; k < kMaxComp; ++k) st.dc[k] = 0;
            end_bit += W * 32;
            decode_subsequence(st, bw, fetch, end_bit, s_tab, sp, sink);
            const int cz = st.c | (st.z << 8);
            if (st.p == s_p[j] && cz == s_cz[j]) flowing = false; // synchronised; still store n / dc
            s_p[j]  = st.p;
            s_n[j]  = st.n;
            s_cz[j] = cz;
#pragma unroll
            for (int k = 0; k < kMaxComp; ++k) s_dc[k][j] = st.dc[k];
        } else {
            flowing = false;
        }
        if (!__syncthreads_or(flowing && j + 1 < lim)) break;
    }

    if (active) {
        const int sub = first_sub + t;
        out.p[sub]    = s_p[t];
        out.n[sub]    = s_n[t];
        out.cz[sub]   = s_cz[t];
        for (int k = 0; k < sp.num_comp; ++k) out.dc[k][sub] = s_dc[k][t];
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: inter-sequence synchronisation
// ------------------------------------------------------------------------------------------------

/// One lane per sequence boundary: carry the exit state of the last subsequence of sequence b-1 into
/// sequence b, b+1, ... until it meets the stored state (or the segment ends). All boundaries advance
/// in lock-step inside ONE workgroup, groups of boundaries are processed in stream order, so a flow
/// that started further upstream always overwrites later and, unlike the
/// reference (Appendix B-4), no pair of boundaries is left unordered. State lives in global memory,
/// bitstream words are read straight from the destuffed buffer.
template <int W>
__global__ __launch_bounds__(1024) void huff_sync_inter(
    const uint32_t* __restrict__ scan32,
    const Segment* __restrict__ segments,
    const int* __restrict__ seg_idx,
    const HuffTableDev* __restrict__ g_tables,
    ScanParams sp,
    SubseqState g)
{
    __shared__ HuffTableDev s_tab[kHuffSlots];
    load_tables(s_tab, g_tables);
    __syncthreads();

    const int num_seq = (sp.num_subseq + T - 1) / T;
    NoSink sink;
    for (int base = 1; base < num_seq; base += blockDim.x) {
        const int b = base + threadIdx.x;
        LaneState st{};
        BitWindow<GlobalFetch> bw{};
        GlobalFetch fetch{nullptr, 0};
        int end_bit  = 0;
        int j        = 0; // global index of the subsequence flowed into next
        int lim      = 0;
        bool flowing = false;
        if (b < num_seq) {
            const int from    = b * T - 1;
            const Segment seg = segments[seg_idx[from]];
            lim               = seg.subseq_offset + seg.subseq_count;
            j                 = from + 1;
            flowing           = j < lim;
            if (flowing) {
                fetch.words     = scan32 + static_cast<size_t>(seg.subseq_offset) * W;
                fetch.seg_words = seg.subseq_count * W;
                st.p            = g.p[from];
                const int cz    = g.cz[from];
                st.c            = cz & 0xFF;
                st.z            = cz >> 8;
                end_bit         = (from - seg.subseq_offset + 1) * (W * 32);
                bw.seek(st.p, fetch);
            }
        }
        while (true) {
            if (flowing && j < lim) {
                st.n = 0;
#pragma unroll
                for (int k = 0; k < kMaxComp; ++k) st.dc[k] = 0;
                end_bit += W * 32;
                decode_subsequence(st, bw, fetch, end_bit, s_tab, sp, sink);
                const int cz = st.c | (st.z << 8);
                if (st.p == g.p[j] && cz == g.cz[j]) flowing = false;
                g.p[j]  = st.p;
                g.n[j]  = st.n;
                g.cz[j] = cz;
                for (int k = 0; k < sp.num_comp; ++k) g.dc[k][j] = st.dc[k];
                ++j;
            } else {
                flowing = false;
            }
            // barrier + workgroup-scope fence: next iteration's reads see this iteration's stores
            if (!__syncthreads_or(flowing && j < lim)) break;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: per-sequence tails
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int block_sum_256(int v, int* s_red)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

/// tails[b] = sum of (n, dc) over the subsequences of sequence b that belong to the segment still
/// open at the end of b. The write pass of sequence b' > b in the same segment adds tails[a..b'-1]
/// (a = sequence holding the segment's first subsequence) to get its offset inside the segment.
__global__ __launch_bounds__(T) void huff_seq_tails(
    const Segment* __restrict__ segments,
    const int* __restrict__ seg_idx,
    ScanParams sp,
    SubseqState g,
    SeqTails tails)
{
    __shared__ int s_red[4];
    const int t         = threadIdx.x;
    const int first_sub = blockIdx.x * T;
    const int nsub      = min(T, sp.num_subseq - first_sub);
    const int last_seg  = seg_idx[first_sub + nsub - 1];
    const int open_from = segments[last_seg].subseq_offset; // global index of that segment's start
    const int sub       = first_sub + t;
    const bool take     = t < nsub && sub >= open_from;
    const int n         = block_sum_256(take ? g.n[sub] : 0, s_red);
    if (t == 0) tails.n[blockIdx.x] = n;
    for (int k = 0; k < sp.num_comp; ++k) {
        const int d = block_sum_256(take ? g.dc[k][sub] : 0, s_red);
        if (t == 0) tails.dc[k][blockIdx.x] = d;
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: write pass
// ------------------------------------------------------------------------------------------------

struct CoefSink {
    static constexpr bool kWrite = true;
    int16_t* out;
    const uint8_t* natural; // LDS copy of the zig-zag -> raster map
    int pos;
    int quota;
    int pred[kMaxComp];
    __device__ __forceinline__ bool full() const { return pos >= quota; }
    __device__ __forceinline__ void store(int v)
    {
        if (pos < quota) out[(pos & ~63) + natural[pos & 63]] = static_cast<int16_t>(v);
    }
    __device__ __forceinline__ void dc(int comp, int diff)
    {
        int v = 0;
#pragma unroll
        for (int k = 0; k < kMaxComp; ++k) {
            pred[k] += comp == k ? diff : 0;
            v = comp == k ? pred[k] : v;
        }
        store(v); // int16 wrap = the reference's int16 prefix sum (decode_dc.cu:129-155)
        ++pos;
    }
    __device__ __forceinline__ void ac(int run, int v)
    {
        pos += run;
        store(v);
        ++pos;
    }
    __device__ __forceinline__ void advance(int k) { pos += k; }
};

/// Exclusive prefix over the 256 lanes of `v` (plain, not segmented), result left in s_scan[0..T].
__device__ __forceinline__ void block_excl_scan_256(int v, int* s_scan, int* s_wave)
{
    const int t       = threadIdx.x;
    const uint32_t in = wave_incl_scan(static_cast<uint32_t>(v));
    __syncthreads(); // previous use of s_scan / s_wave is over
    if (lane_id() == 63) s_wave[t >> 6] = static_cast<int>(in);
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) off += k < (t >> 6) ? s_wave[k] : 0;
    s_scan[t] = static_cast<int>(in) - v + off;
    if (t == T - 1) s_scan[T] = static_cast<int>(in) + off;
    __syncthreads();
}

/// Re-decode every subsequence from its predecessor's synchronised exit state and store the
/// non-zero coefficients in stream order (data unit after data unit, natural order inside, DC
/// already absolute). Output position of subsequence i inside its segment = sum of n over the
/// segment's earlier subsequences: in-sequence part by an LDS scan, earlier sequences via tails.
template <int W>
__global__ __launch_bounds__(T) void huff_write(
    const uint32_t* __restrict__ scan32,
    const Segment* __restrict__ segments,
    const int* __restrict__ seg_idx,
    const HuffTableDev* __restrict__ g_tables,
    ScanParams sp,
    SubseqState g,
    SeqTails tails,
    int16_t* __restrict__ coef)
{
    __shared__ uint32_t s_img[SeqImage<W>::kWords];
    __shared__ uint32_t s_tail[3];
    __shared__ HuffTableDev s_tab[kHuffSlots];
    __shared__ int s_scan[T + 1];
    __shared__ int s_wave[4];
    __shared__ int s_carry[1 + kMaxComp];
    __shared__ uint8_t s_nat[64];

    const int t         = threadIdx.x;
    const int first_sub = blockIdx.x * T;
    const int nsub      = min(T, sp.num_subseq - first_sub);

    load_tables(s_tab, g_tables);
    load_sequence<W>(s_img, s_tail, scan32, first_sub, nsub, sp.num_subseq);
    if (t < 64) {
        constexpr uint8_t nat[64] = JG_ORDER_NATURAL;
        s_nat[t]                  = nat[t];
    }

    // carry-in of the segment that is open at the sequence's first subsequence
    {
        const Segment seg0 = segments[seg_idx[first_sub]];
        const int a        = seg0.subseq_offset / T; // sequence holding the segment's start
        int cn = 0, cd[kMaxComp] = {0, 0, 0, 0};
        if (seg0.subseq_offset < first_sub) {
            for (int b = a + t; b < static_cast<int>(blockIdx.x); b += T) {
                cn += tails.n[b];
                for (int k = 0; k < sp.num_comp; ++k) cd[k] += tails.dc[k][b];
            }
        }
        cn = block_sum_256(cn, s_wave);
        if (t == 0) s_carry[0] = cn;
        for (int k = 0; k < sp.num_comp; ++k) {
            const int d = block_sum_256(cd[k], s_wave);
            if (t == 0) s_carry[1 + k] = d;
        }
    }
    __syncthreads();

    const bool active = t < nsub;
    const int sub     = first_sub + t;
    int seg_i = 0, rel = 0, ts = 0; // ts = local index of the segment's first subsequence (clamped to 0)
    Segment seg{0, 0};
    bool carried = false; // segment started before this sequence
    if (active) {
        seg_i   = seg_idx[sub];
        seg     = segments[seg_i];
        rel     = sub - seg.subseq_offset;
        carried = seg.subseq_offset < first_sub;
        ts      = carried ? 0 : seg.subseq_offset - first_sub;
    }

    CoefSink sink;
    sink.out     = coef;
    sink.natural = s_nat;
    int nprefix  = 0;
    {
        block_excl_scan_256(active ? g.n[sub] : 0, s_scan, s_wave);
        nprefix = s_scan[t] - s_scan[ts] + (carried ? s_carry[0] : 0);
#pragma unroll
        for (int k = 0; k < kMaxComp; ++k) {
            sink.pred[k] = 0;
            if (k < sp.num_comp) {
                block_excl_scan_256(active ? g.dc[k][sub] : 0, s_scan, s_wave);
                sink.pred[k] = s_scan[t] - s_scan[ts] + (carried ? s_carry[1 + k] : 0);
            }
        }
    }
    if (!active) return;

    const int du_words  = sp.du_per_mcu * 64;
    const int seg_mcus0 = seg_i * sp.mcus_per_segment;
    const int seg_mcus1 = min(seg_mcus0 + sp.mcus_per_segment, sp.total_mcus); // Appendix B-5 clamp
    sink.pos            = seg_mcus0 * du_words + nprefix;
    sink.quota          = seg_mcus1 * du_words;

    LaneState st{};
    if (rel > 0) {
        st.p         = g.p[sub - 1];
        const int cz = g.cz[sub - 1];
        st.c         = cz & 0xFF;
        st.z         = cz >> 8;
    }
    LdsFetch<W> fetch{s_img, s_tail, (seg.subseq_offset - first_sub) * W, seg.subseq_count * W};
    BitWindow<LdsFetch<W>> bw{};
    bw.seek(st.p, fetch);
    decode_subsequence(st, bw, fetch, (rel + 1) * (W * 32), s_tab, sp, sink);
}

// ------------------------------------------------------------------------------------------------
// dequantisation + inverse DCT
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int unfixh(int x) { return static_cast<int16_t>((x + 0x8000) >> 16); }
__device__ __forceinline__ int unfixo(int x) { return (x + 0x1000) >> 13; }

/// 8-point fixed-point inverse DCT, the arithmetic of the reference's `idct_vector`
/// (src/idct.cu:49-95): Q15 even part, Q13 odd part, results rounded to int16.
__device__ __forceinline__ void idct8(int (&v)[8])
{
    constexpr int cos_1_4 = 0x5a82, sin_1_8 = 0x30fc, cos_1_8 = 0x7642;
    constexpr int osin_1_16 = 0x063e, osin_5_16 = 0x1a9b, ocos_1_16 = 0x1f63, ocos_5_16 = 0x11c7;

    const int e0 = (v[0] + v[4]) * cos_1_4;
    const int e1 = (v[0] - v[4]) * cos_1_4;
    const int e2 = v[2] * sin_1_8 - v[6] * cos_1_8;
    const int e3 = v[6] * sin_1_8 + v[2] * cos_1_8;
    const int a0 = e0 + e3, a1 = e1 + e2, a2 = e1 - e2, a3 = e0 - e3;

    const int m0 = unfixo((v[3] + v[5]) * cos_1_4);
    const int m1 = unfixo((v[3] - v[5]) * cos_1_4);
    const int q1 = v[1] << 2, q7 = v[7] << 2;
    const int o0 = q1 + m0, o1 = q7 + m1, o2 = q1 - m0, o3 = q7 - m1;
    const int b0 = o0 * ocos_1_16 + o1 * osin_1_16;
    const int b1 = o0 * osin_1_16 - o1 * ocos_1_16;
    const int b2 = o2 * ocos_5_16 + o3 * osin_5_16;
    const int b3 = o2 * osin_5_16 - o3 * ocos_5_16;

    v[0] = unfixh(a0 + b0);
    v[1] = unfixh(a1 + b3);
    v[2] = unfixh(a2 + b2);
    v[3] = unfixh(a3 + b1);
    v[4] = unfixh(a3 - b1);
    v[5] = unfixh(a2 - b2);
    v[6] = unfixh(a1 - b3);
    v[7] = unfixh(a0 - b0);
}

constexpr int kIdctDuPerBlock = 32;        // 8 lanes per data unit, 256 lanes
constexpr int kIdctRowStride  = 8 + 2;     // int16 per staged row (+2: column reads spread over banks)

/// One data unit per 8 lanes, read straight from the stream-order coefficient buffer (16 bytes =
/// one row per lane, fully coalesced). Steps and int16 truncation points are those of the reference
/// `idct_kernel` (src/idct.cu:146-223): (int16)(coef * q) -> column pass -> row pass -> +128 -> clamp.
/// The MCU geometry (reference decode_transpose.cu:65-131) is applied when the 8x8 pixels are stored.
__global__ __launch_bounds__(256) void idct_kernel(
    const int16_t* __restrict__ coef, const uint8_t* __restrict__ qtables, IdctParams ip)
{
    __shared__ int16_t s_blk[kIdctDuPerBlock][8][kIdctRowStride];

    const int t   = threadIdx.x;
    const int r   = t & 7;  // row (passes 1 and 3) or column (pass 2) handled by this lane
    const int dl  = t >> 3; // data unit inside the workgroup
    const int du  = blockIdx.x * kIdctDuPerBlock + dl;
    const bool in = du < ip.num_du;

    const int mcu = du / ip.du_per_mcu;
    const int k   = du - mcu * ip.du_per_mcu;
    const int sc  = in ? ip.du_comp[k] : 0;

    int v[8];
    if (in) {
        const uint4 raw = *reinterpret_cast<const uint4*>(coef + static_cast<size_t>(du) * 64 + r * 8);
        const uint2 qr  = *reinterpret_cast<const uint2*>(qtables + ip.qidx[sc] * 64 + r * 8);
        const uint32_t cw[4] = {raw.x, raw.y, raw.z, raw.w};
        const uint32_t qw[2] = {qr.x, qr.y};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = static_cast<int16_t>(cw[i >> 1] >> (16 * (i & 1)));
            const int q = (qw[i >> 2] >> (8 * (i & 3))) & 0xFF; // unsigned (Appendix B-3)
            s_blk[dl][r][i] = static_cast<int16_t>(c * q);
        }
    }
    __syncthreads();
    if (in) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = s_blk[dl][i][r];
        idct8(v);
#pragma unroll
        for (int i = 0; i < 8; ++i) s_blk[dl][i][r] = static_cast<int16_t>(v[i]);
    }
    __syncthreads();
    if (!in) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = s_blk[dl][r][i];
    idct8(v);

    const int mx = mcu % ip.mcus_x, my = mcu / ip.mcus_x;
    const int x0 = (mx * ip.comp_h[sc] + ip.du_dx[k]) * 8;
    const int y  = (my * ip.comp_v[sc] + ip.du_dy[k]) * 8 + r;
    if (y >= ip.size_y[sc] || x0 >= ip.size_x[sc]) return;
    uint32_t px[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = static_cast<int16_t>(v[i] + 128);
        px[i]       = static_cast<uint32_t>(min(max(s, 0), 255));
    }
    uint8_t* row = ip.plane[sc] + static_cast<size_t>(y) * ip.pitch[sc] + x0;
    if (x0 + 8 <= ip.size_x[sc] && (reinterpret_cast<uintptr_t>(row) & 7) == 0) {
        uint2 o;
        o.x = px[0] | px[1] << 8 | px[2] << 16 | px[3] << 24;
        o.y = px[4] | px[5] << 8 | px[6] << 16 | px[7] << 24;
        *reinterpret_cast<uint2*>(row) = o;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (x0 + i < ip.size_x[sc]) row[i] = static_cast<uint8_t>(px[i]);
        }
    }
}

/// Chroma replication: each lane produces 4 consecutive output pixels of one row.
__global__ __launch_bounds__(256) void upsample_kernel(
    const uint8_t* __restrict__ src, int src_pitch, int src_w, int src_h,
    uint8_t* __restrict__ dst, int dst_pitch, int dst_w, int dst_h,
    int num_x, int den_x, int num_y, int den_y)
{
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y  = blockIdx.y;
    if (x0 >= dst_w || y >= dst_h) return;
    const int sy        = min(y * num_y / den_y, src_h - 1);
    const uint8_t* srow = src + static_cast<size_t>(sy) * src_pitch;
    uint8_t* drow       = dst + static_cast<size_t>(y) * dst_pitch + x0;
    uint32_t px[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) px[i] = srow[min((x0 + i) * num_x / den_x, src_w - 1)];
    if (x0 + 4 <= dst_w && (reinterpret_cast<uintptr_t>(drow) & 3) == 0) {
        *reinterpret_cast<uint32_t*>(drow) = px[0] | px[1] << 8 | px[2] << 16 | px[3] << 24;
    } else {
        for (int i = 0; i < 4 && x0 + i < dst_w; ++i) drow[i] = static_cast<uint8_t>(px[i]);
    }
}

template <int W>
hipError_t launch_huffman_w(
    HuffStage which,
    const uint32_t* scan32,
    const Segment* d_segments,
    const int* d_seg_idx,
    const HuffTableDev* d_tables,
    const ScanParams& sp,
    SubseqState st,
    SeqTails tails,
    int16_t* d_coef,
    hipStream_t stream)
{
    const int num_seq = (sp.num_subseq + T - 1) / T;
    switch (which) {
    case kHuffSyncIntra:
        huff_sync_intra<W><<<num_seq, T, 0, stream>>>(scan32, d_segments, d_seg_idx, d_tables, sp, st);
        break;
    case kHuffSyncInter:
        if (num_seq > 1) {
            const int want  = ((num_seq - 1 + 63) / 64) * 64;
            const int lanes = want < 1024 ? want : 1024;
            huff_sync_inter<W><<<1, lanes, 0, stream>>>(scan32, d_segments, d_seg_idx, d_tables, sp, st);
        }
        break;
    case kHuffTails:
        huff_seq_tails<<<num_seq, T, 0, stream>>>(d_segments, d_seg_idx, sp, st, tails);
        break;
    case kHuffWrite:
        huff_write<W><<<num_seq, T, 0, stream>>>(scan32, d_segments, d_seg_idx, d_tables, sp, st, tails, d_coef);
        break;
    }
    return hipGetLastError();
}

} // namespace

bool subseq_bytes_supported(int b) { return b == 32 || b == 64 || b == 128; }

hipError_t launch_destuff(
    const uint8_t* d_bytes,
    uint8_t* d_destuffed,
    int* d_seg_idx,
    const DestuffChunk* d_chunks,
    int num_chunks,
    int subseq_bytes,
    hipStream_t stream)
{
    if (num_chunks == 0) return hipSuccess;
    int shift = 0;
    while ((1 << shift) < subseq_bytes) ++shift;
    destuff_kernel<<<num_chunks, 256, 0, stream>>>(d_bytes, d_destuffed, d_seg_idx, d_chunks, shift);
    return hipGetLastError();
}

hipError_t launch_huffman_stage(
    HuffStage which,
    const uint8_t* d_destuffed,
    const Segment* d_segments,
    const int* d_seg_idx,
    const HuffTableDev* d_tables,
    const ScanParams& sp,
    SubseqState st,
    SeqTails tails,
    int16_t* d_coef,
    hipStream_t stream)
{
    if (sp.num_subseq == 0) return hipSuccess;
    const uint32_t* scan32 = reinterpret_cast<const uint32_t*>(d_destuffed);
    switch (sp.subseq_words) {
    case 8: return launch_huffman_w<8>(which, scan32, d_segments, d_seg_idx, d_tables, sp, st, tails, d_coef, stream);
    case 16: return launch_huffman_w<16>(which, scan32, d_segments, d_seg_idx, d_tables, sp, st, tails, d_coef, stream);
    case 32: return launch_huffman_w<32>(which, scan32, d_segments, d_seg_idx, d_tables, sp, st, tails, d_coef, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_idct(
    const int16_t* d_coef, const uint8_t* d_qtables, const IdctParams& ip, hipStream_t stream)
{
    if (ip.num_du == 0) return hipSuccess;
    const int blocks = (ip.num_du + kIdctDuPerBlock - 1) / kIdctDuPerBlock;
    idct_kernel<<<blocks, 256, 0, stream>>>(d_coef, d_qtables, ip);
    return hipGetLastError();
}

hipError_t launch_upsample(
    const uint8_t* src, int src_pitch, int src_w, int src_h,
    uint8_t* dst, int dst_pitch, int dst_w, int dst_h,
    int num_x, int den_x, int num_y, int den_y, hipStream_t stream)
{
    if (dst_w <= 0 || dst_h <= 0) return hipSuccess;
    const dim3 grid((dst_w + 1023) / 1024, dst_h);
    upsample_kernel<<<grid, 256, 0, stream>>>(
        src, src_pitch, src_w, src_h, dst, dst_pitch, dst_w, dst_h, num_x, den_x, num_y, den_y);
    return hipGetLastError();
}

} // namespace jg
