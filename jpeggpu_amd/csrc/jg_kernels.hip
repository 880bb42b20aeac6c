// jg_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the baseline JPEG decode path.
//
//   destuff_kernel          byte-stuffing / restart-marker removal   (reference src/decode_destuff.cu:37-361)
//   huff_sync_intra         speculative decode + intra-sequence sync (reference decode_huffman.cu:413-524)
//   huff_sync_tail          inter-sequence sync + unfinished flows   (reference decode_huffman.cu:534-621)
//   huff_seq_tails          per-sequence sums of n / DC              (replaces cub ExclusiveScanByKey :818-869
//                                                                     and the DC scans of decode_dc.cu:88-169)
//   huff_write              final decode -> symbol stream, absolute DC (reference decode_huffman.cu:627-682)
//   idct_kernel             gather + dequant + 8x8 fixed-point IDCT in stream order
//                                                                    (reference idct.cu:44-223 + decode_transpose.cu:41-132)
//
// Everything is integer / bit-serial: no MFMA. Every kernel takes a job source: one ScanJob by
// value (drop-in API) or an array indexed by blockIdx.y (batch API: one launch per stage for many
// images, which is what fills 256 CUs). The destuffed bitstream is kept in tiles of 32 subsequences,
// word-major (jg_defs.h), so that the 64 lanes of a wave, each walking its own subsequence, share cache
// lines; nothing but the Huffman tables and a small write-combining ring lives in LDS.
#define JG_TABS_IN_LDS 1 // table offsets handed to the symbol loop are absolute LDS addresses (jg_huff_core.h)
#include "jg_huff_core.h"
#include "jg_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "jg_bytes.h"
#include "jg_wave.h"

namespace jg {

namespace {

constexpr int T   = kSeqLanes;   // lanes per workgroup in the Huffman kernels
// Subsequences a workgroup owns (SEQ) and lanes that re-decode the tail of the previous sequence (OV = T - SEQ): per job,
// ScanParams::seq_subseq -- 240 + 16 for a lone decode, 255 + 1 in batches (jg_defs.h). Every kernel reads it into a local
// `SEQ` (uniform: a scalar register).
constexpr int kMaxSeq = kSeqLanes - 1;

/// Pointers read out of a job that lives in memory are generic to the compiler, and generic (flat) loads
/// and stores count against lgkmcnt as well as vmcnt: every wait for an LDS table read would also wait for
/// the bitstream word prefetched one refill ahead. The kernels therefore work on a view of the job whose
/// pointers are qualified as global memory (kernel arguments passed by value are inferred global anyway).
#define JG_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ JG_GLOBAL T* as_global(T* p)
{
    return (JG_GLOBAL T*)p;
}

/// Class types do not copy through address-space-qualified pointers: move them as vectors of dwords.
template <class T>
__device__ __forceinline__ T ld_global(JG_GLOBAL const T* p)
{
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "dword-sized objects only");
    typedef uint32_t V __attribute__((ext_vector_type(sizeof(T) / 4)));
    const V v = *reinterpret_cast<JG_GLOBAL const V*>(p);
    return __builtin_bit_cast(T, v);
}
template <class T>
__device__ __forceinline__ void st_global(JG_GLOBAL T* p, const T& value)
{
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "dword-sized objects only");
    typedef uint32_t V __attribute__((ext_vector_type(sizeof(T) / 4)));
    *reinterpret_cast<JG_GLOBAL V*>(p) = __builtin_bit_cast(V, value);
}

struct JobView {
    JG_GLOBAL const uint8_t* bytes;
    JG_GLOBAL const DestuffChunk* chunks;
    JG_GLOBAL const Segment* segments;
    JG_GLOBAL const uint8_t* tables;
    JG_GLOBAL const uint8_t* tables_sync;
    JG_GLOBAL const uint16_t* qtables;
    JG_GLOBAL uint8_t* destuffed;
    JG_GLOBAL int* seg_idx;
    JG_GLOBAL int* st_p;
    JG_GLOBAL int* st_n;
    JG_GLOBAL int* st_cz;
    JG_GLOBAL uint32_t* st_dc01;
    JG_GLOBAL uint32_t* st_dc23;
    JG_GLOBAL uint8_t* pending;
    JG_GLOBAL int* bnd_p;
    JG_GLOBAL int* bnd_cz;
    JG_GLOBAL int* flow_list;
    JG_GLOBAL const int* tail_parts;
    int num_tail_parts;
    JG_GLOBAL int* tails_n;
    JG_GLOBAL uint32_t* tails_dc01;
    JG_GLOBAL uint32_t* tails_dc23;
    JG_GLOBAL int* mh_p;
    JG_GLOBAL int* mh_cz;
    JG_GLOBAL uint32_t* mh_link;
    JG_GLOBAL uint2_t* mh_pool;
    JG_GLOBAL uint8_t* mh_known;
    JG_GLOBAL const MhBlock* mh_blocks;
    JG_GLOBAL uint16_t* mh_blk_exit;
    JG_GLOBAL uint16_t* mh_blk_entry;
    int num_mh_blocks;
    JG_GLOBAL uint16_t* sym;
    JG_GLOBAL uint2_t* du_tab;
    uint32_t sym_region;
    uint64_t sym_entries;
    int num_chunks;
    int num_seq;
    const ScanParams& sp;
    const IdctParams& ip;
    __device__ __forceinline__ explicit JobView(const ScanJob& j)
        : bytes(as_global(j.bytes)), chunks(as_global(j.chunks)), segments(as_global(j.segments)),
          tables(as_global(j.tables)), tables_sync(as_global(j.tables_sync)), qtables(as_global(j.qtables)), destuffed(as_global(j.destuffed)),
          seg_idx(as_global(j.seg_idx)), st_p(as_global(j.st_p)), st_n(as_global(j.st_n)), st_cz(as_global(j.st_cz)),
          st_dc01(as_global(j.st_dc01)), st_dc23(as_global(j.st_dc23)), pending(as_global(j.pending)),
          bnd_p(as_global(j.bnd_p)), bnd_cz(as_global(j.bnd_cz)), flow_list(as_global(j.flow_list)), tail_parts(as_global(j.tail_parts)), num_tail_parts(j.num_tail_parts),
          tails_n(as_global(j.tails_n)), tails_dc01(as_global(j.tails_dc01)), tails_dc23(as_global(j.tails_dc23)),
          mh_p(as_global(j.mh_p)), mh_cz(as_global(j.mh_cz)), mh_link(as_global(j.mh_link)), mh_pool(as_global(j.mh_pool)), mh_known(as_global(j.mh_known)),
          mh_blocks(as_global(j.mh_blocks)), mh_blk_exit(as_global(j.mh_blk_exit)), mh_blk_entry(as_global(j.mh_blk_entry)), num_mh_blocks(j.num_mh_blocks),
          sym(as_global(j.sym)), du_tab(as_global(j.du_tab)), sym_region(j.sym_region), sym_entries(j.sym_entries),
          num_chunks(j.num_chunks), num_seq(j.num_seq), sp(j.sp), ip(j.ip)
    {
    }
};

struct JobByValue {
    // One image: ~14 dependent flow iterations decide the time, the speculative pass is one of them.
    static constexpr bool kSpeculateStateOnly = false;
    static constexpr bool kRepackFlows = true; // huff_sync_intra: flows that outlive the first iteration are packed into the lowest lanes
    ScanJob job;
    __device__ __forceinline__ const ScanJob& get() const { return job; }
};
struct JobsByValue {
    // All scans of ONE image (a file with several scans decoded on its own): one launch per stage, a scan per
    // blockIdx.y -- the scans are independent, and a 39 MP file of three scans spends a third of the time of three
    // launch sequences one after the other.
    static constexpr bool kSpeculateStateOnly = false;
    static constexpr bool kRepackFlows = true;
    ScanJob jobs[kMaxScans];
    __device__ __forceinline__ const ScanJob& get() const { return jobs[blockIdx.y]; }
};
struct JobArray {
    // Batches run one flow iteration, so the speculative pass is half of the sequence kernel's work:
    // it tracks the exit state only (-10 % kernel time); single-image latency is 4 % better without.
    static constexpr bool kSpeculateStateOnly = true;
    // The sequence kernel of a batch is bound by how many workgroups a CU holds (LDS: the sync table pack): the 4 KB the
    // re-packing needs cost one in five (538 -> 680 us per 64 images), and follow-up iterations inside that kernel cost
    // more than the tail kernel's trips they replace (jg_decoder.cpp, sync_iters): batches run huff_sync_intra_batch.
    static constexpr bool kRepackFlows = false;
    const ScanJob* jobs;
    __device__ __forceinline__ const ScanJob& get() const { return jobs[blockIdx.y]; }
};

struct JobArrayLow {
    // A batch that does not fill the chip (jpeggpu_ext_decode_batch with a handful of images): what such a call waits for is
    // the chain of dependent flow iterations, as a lone decode does, and LDS is not what bounds the sequence kernel then: the
    // lone decode's kernel (all flows kept in the workgroup, survivors re-packed), one job per blockIdx.y. Only
    // huff_sync_intra is instantiated for it; every other stage of such a batch runs with JobArray.
    static constexpr bool kSpeculateStateOnly = false;
    static constexpr bool kRepackFlows = true;
    const ScanJob* jobs;
    __device__ __forceinline__ const ScanJob& get() const { return jobs[blockIdx.y]; }
};

struct JobSingle {
    // One job that lives in device memory, whatever blockIdx.y is: the lone decode of a device-scanned image (the
    // second dimension of the multi-hypothesis kernels' grid is the hypothesis).
    static constexpr bool kSpeculateStateOnly = false;
    static constexpr bool kRepackFlows = true;
    const ScanJob* job;
    __device__ __forceinline__ const ScanJob& get() const { return *job; }
};

// ------------------------------------------------------------------------------------------------
// destuff
// ------------------------------------------------------------------------------------------------

/// One aligned 4 KiB window of the transferred bytes clipped to one segment = one chunk; a workgroup takes TWO
/// neighbouring chunks, with the loads of both issued before the first is worked on: a chunk is little work behind
/// two dependent loads (its record, then its bytes), so a workgroup's life was mostly those two latencies.
/// Lane t owns 16 consecutive source bytes. Byte rule (reference src/decode_destuff.cu:37-44): a byte is
/// data iff (prev == FF and b == 00) or (prev != FF and b != FF); the first case stores FF.
/// The compacted bytes are staged in LDS at the destination's 16-byte phase and leave as whole
/// 16-byte stores except at the two ragged ends (neighbouring chunks own the other bytes there).
constexpr int kDestuffChunksPerWg = 2;

template <class J_t>
__device__ __forceinline__ void destuff_chunk(const J_t& J, const DestuffChunk& ck, const uint4& v, uint32_t prev_of_lane0, uint8_t* s_out, uint32_t* s_wave)
{
    JG_GLOBAL uint8_t* __restrict__ dst = J.destuffed;
    const int t         = threadIdx.x;
    const uint32_t gpos = ck.win_off + t * 16;
    uint32_t w[4]       = {v.x, v.y, v.z, v.w};
    uint32_t prev       = __shfl_up(w[3] >> 24, 1);
    if (lane_id() == 0) prev = prev_of_lane0;

    // Byte rule on four bytes at a time (jg_bytes.h): F / Z = bytes equal to FF / 00, PF = the byte in front is
    // FF. A stuffed zero is data and stands for the FF in front of it: the word is patched so that the
    // compaction below stores plain bytes.
    uint32_t F[4], Z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        F[j] = bytes_ff(w[j]);
        Z[j] = bytes_zero(w[j]);
    }
    uint32_t f_before = prev == 0xFFu ? 0x80000000u : 0u;
    if (ck.first) { // the byte in front of a segment's first byte belongs to a marker: never stuffing
        const uint32_t d = ck.begin - gpos;
        if (d == 0) f_before = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (d - 1 - 4 * j < 4u) F[j] &= ~(0x80u << (8 * ((d - 1) & 3u))); // that byte lies outside [begin, end)
    }
    uint32_t mask = 0; // bit i: byte i is data
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t PF      = of_previous_byte(F[j], j ? F[j - 1] : f_before);
        const uint32_t stuffed = PF & Z[j];
        const uint32_t data    = stuffed | (~(PF | F[j]) & kHi80);
        w[j] |= spread80(stuffed);
        mask |= collapse80(data) << (4 * j);
    }
    {
        const uint32_t lo = min(max(static_cast<int>(ck.begin - gpos), 0), 16), hi = min(max(static_cast<int>(ck.end - gpos), 0), 16);
        mask &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    }

    const uint32_t cnt  = __popc(mask);
    const uint32_t incl = wave_incl_scan(cnt);
    if (lane_id() == 63) s_wave[t >> 6] = incl;
    __syncthreads();
    uint32_t off   = incl - cnt;
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
                s_out[o++] = static_cast<uint8_t>(w[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
    __syncthreads();

    // write-out into the tiled layout (jg_defs.h): 4-byte words, byte stores at the two ragged ends
    // (neighbouring chunks own the other bytes of those words). A word also goes into the slots of the
    // neighbouring rows that mirror it: the first two words of a subsequence behind the previous row, the last
    // one in front of the next row -- whoever stores a word stores all its copies, so no slot has two writers.
    const int log2w          = 31 - __clz(J.sp.subseq_words);
    const uint32_t wmask     = static_cast<uint32_t>(J.sp.subseq_words) - 1u;
    JG_GLOBAL uint32_t* const dst32 = reinterpret_cast<JG_GLOBAL uint32_t*>(dst);
    // up to three slots (32-bit word indices) of linear word lw; returns how many
    const auto slots_of = [&](uint32_t lw, uint32_t (&at)[3]) -> int {
        const uint32_t row = lw >> log2w, k = lw & wmask;
        int n   = 0;
        at[n++] = tiled_slot(row, k + kRowLeadWords, log2w);
        if (k < static_cast<uint32_t>(kRowTailWords) && row > 0) at[n++] = tiled_slot(row - 1, wmask + 1u + kRowLeadWords + k, log2w);
        if (k == wmask) at[n++] = tiled_slot(row + 1, 0, log2w); // the row behind the last subsequence is spare space
        return n;
    };
    const uint32_t word0     = (ck.dst_off - phase) >> 2; // linear word of s_out[0]
    const uint32_t lo = phase, hi = phase + total;        // valid LDS byte range
    const auto word_at = [&](uint32_t lw) -> uint32_t { // linear word lw of the scan, most significant byte first
        return __builtin_bswap32(*reinterpret_cast<const uint32_t*>(s_out + (lw - word0) * 4u));
    };
    // The words that lie wholly inside the chunk, [wa, wb): row R of the tiled buffer holds the linear words
    // R * W - 1 .. R * W + W + 1 in its slots 0 .. W + 2 (jg_defs.h: own words and the mirrored neighbours), and the
    // same slot of four consecutive rows is 16 consecutive bytes of a tile. So the write-out walks PIECES (tile, slot,
    // group of four rows): four LDS words W apart, one 16-byte store, and the 64 lanes of a wave fill 1 KiB of
    // consecutive memory. Stored word by word in stream order a wave's store touched 64 lines with 4 bytes each:
    // half the kernel's time. Pieces at the chunk's first and last rows, which a neighbouring chunk shares, fall
    // back to single words.
    const uint32_t wa = word0 + ((lo + 3u) >> 2), wb = word0 + (hi >> 2);
    if (wa < wb) {
        const int lr          = tile_rows_log2(log2w);
        const uint32_t W      = wmask + 1u;
        const uint32_t gmask  = (1u << (lr - 2)) - 1u;                  // groups of four rows in a tile, minus one
        const uint32_t pieces = (W + kRowExtraWords) << (lr - 2);       // of one tile
        const uint32_t r_min  = wa >= 2u ? (wa - 2u) >> log2w : 0u;     // first and last row that hold one of the words
        const uint32_t r_max  = wb >> log2w;
        for (uint32_t tile = r_min >> lr; tile <= r_max >> lr; ++tile) {
            for (uint32_t p = t; p < pieces; p += 256) {
                const uint32_t slot = p >> (lr - 2), j = p & gmask;
                const uint32_t row0 = (tile << lr) + 4u * j;
                const uint32_t lw0  = (row0 << log2w) + slot - 1u;      // row 0, slot 0 wraps around: outside [wa, wb)
                bool in[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) in[i] = lw0 + static_cast<uint32_t>(i) * W - wa < wb - wa;
                JG_GLOBAL uint32_t* at = dst32 + (((tile * (W + kRowExtraWords) + slot) << lr) + 4u * j);
                if (in[0] && in[1] && in[2] && in[3]) {
                    st_global(reinterpret_cast<JG_GLOBAL uint4*>(at), make_uint4(word_at(lw0), word_at(lw0 + W), word_at(lw0 + 2u * W), word_at(lw0 + 3u * W)));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (in[i]) at[i] = word_at(lw0 + static_cast<uint32_t>(i) * W);
                }
            }
        }
    }
    // the chunk's first and last word when a neighbouring chunk owns the other bytes: byte stores into every copy
    if (t < 2 && lo < hi) {
        const uint32_t f = lo >> 2, l = hi >> 2; // LDS words that hold the first byte and the one behind the last
        const bool mine  = t == 0 ? (lo & 3u) != 0 : (hi & 3u) != 0 && (l != f || (lo & 3u) == 0);
        const uint32_t b0 = (t == 0 ? f : l) * 4u;
        if (mine) {
            uint32_t at[3];
            const int n = slots_of(word0 + (b0 >> 2), at);
            for (uint32_t b = b0 > lo ? b0 : lo; b < b0 + 4 && b < hi; ++b)
                for (int i = 0; i < n; ++i) dst[at[i] * 4 + (3u - (b & 3))] = s_out[b];
        }
    }
    // zero the tail of the segment up to its subsequence-aligned end
    if (ck.pad_end) {
        for (uint32_t b = ck.dst_off + total + t; b < ck.pad_end; b += 256) {
            uint32_t at[3];
            const int n = slots_of(b >> 2, at);
            for (int i = 0; i < n; ++i) dst[at[i] * 4 + (3u - (b & 3))] = 0;
        }
    }
    // subsequences that start inside this chunk's destination range belong to this segment
    if (total) {
        const uint32_t sb    = static_cast<uint32_t>(J.sp.subseq_words) * 4u;
        const uint32_t first = (ck.dst_off + sb - 1) / sb;
        for (uint32_t s = first + t; s * sb < ck.dst_off + total; s += 256) J.seg_idx[s] = ck.seg;
    }
}

template <class JS>
__global__ __launch_bounds__(256) void destuff_kernel(JS js)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_out[kDestuffWin + 32];
    __shared__ uint32_t s_wave[4];

    const JobView J(js.get());
    const int c0 = static_cast<int>(blockIdx.x) * kDestuffChunksPerWg;
    if (c0 >= J.num_chunks) return;
    const int n = min(kDestuffChunksPerWg, J.num_chunks - c0);
    JG_GLOBAL const uint8_t* __restrict__ src = J.bytes;
    const int t = threadIdx.x;
    DestuffChunk ck[kDestuffChunksPerWg];
    uint4 v[kDestuffChunksPerWg];
    uint32_t prev0[kDestuffChunksPerWg];
#pragma unroll
    for (int k = 0; k < kDestuffChunksPerWg; ++k) ck[k] = ld_global(J.chunks + c0 + (k < n ? k : 0));
#pragma unroll
    for (int k = 0; k < kDestuffChunksPerWg; ++k) {
        const uint32_t gpos = ck[k].win_off + t * 16;
        v[k]                = ld_global(reinterpret_cast<JG_GLOBAL const uint4*>(src + gpos));
        prev0[k]            = 0u;
        if (lane_id() == 0 && gpos > 0) prev0[k] = src[gpos - 1]; // the byte in front of a wave's first one
    }
#pragma unroll
    for (int k = 0; k < kDestuffChunksPerWg; ++k) {
        if (k < n) { // uniform
            if (k) __syncthreads(); // the staging buffer is free again
            destuff_chunk(J, ck[k], v[k], prev0[k], s_out, s_wave);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: bitstream access
// ------------------------------------------------------------------------------------------------

/// Words of the tiled destuffed buffer (jg_defs.h); destuff_kernel stores every word most significant byte
/// first, so a loaded word goes into the bit window as it is, one refill after its load was issued. The lanes of
/// a wave walk neighbouring subsequences at about the same pace, so their refills share 128-byte lines and hit
/// L1; nothing is staged in LDS, which keeps the workgroups of the sync kernel on a CU many.
///
/// A decode of subsequence t works in ROW t of the tiled buffer (set_row), which also mirrors the neighbouring
/// subsequences' words it can touch (jg_defs.h): the window starts at most 31 bits in front of the subsequence
/// (slot 0) and a state-only pass looks at most one 32-bit peek plus one prefetched word past its end (slots
/// W + 1, W + 2; one more, never used, is loaded from whatever follows the row). The position is therefore ONE
/// byte offset and the step to the next word is one add (128 bytes; 64 in the 16-row tiles of W = 64). Only the write pass, whose lanes run on until the data
/// unit they started is complete, can leave the row (kCrossRows): it then continues at the third word of the next
/// row, a rare branch.
///
/// Reads are not clamped into the segment and nothing is zeroed behind its end (the reference's reader does
/// both, decode_huffman_reader.hpp:110-152). A lane never commits a symbol that uses a bit past the end of its
/// segment on a valid stream -- whether a symbol still fits is decided by the bits in front of it (prefix
/// code), and the write pass stops at the segment's data-unit quota -- so what lies behind the end cannot
/// change p, c, z, n, the DC sums or an emitted coefficient. On a corrupt stream a lane of the write pass may
/// run up to one data unit (64 symbols of at most 27 bits: 54 words, plus the window's 3) past its
/// subsequence: the buffer carries more than one spare tile of rows behind the last one (jg_decoder.cpp),
/// whatever they hold.
template <int W, bool kCrossRows = false>
struct GlobalFetch {
    static constexpr int kLog2W = W == 8 ? 3 : W == 16 ? 4 : W == 32 ? 5 : 6;
    static_assert((1 << kLog2W) == W, "subsequence words must be 8, 16, 32 or 64");
    static constexpr int kLog2Rows      = tile_rows_log2(kLog2W);          // rows per tile: 32, or 16 at W = 64 (jg_defs.h)
    static constexpr uint32_t kSlotBytes = 4u << kLog2Rows;                 // one slot of every row of a tile
    static constexpr uint32_t kRowBytes  = kSlotBytes * (W + kRowExtraWords); // slot 0 of a row to slot 0 of the same row of the next tile
    JG_GLOBAL const uint32_t* scan32; // the scan's destuffed buffer (tiled): the same for every lane (scalar base address)
    uint32_t row0;                    // byte offset of slot 0 of the row the decode works in
    int row_word0;                    // segment-relative index of that row's first own word (slot 1)
    struct Pos {
        uint32_t off; // byte offset of the word in the tiled buffer
        uint32_t end; // kCrossRows: offset one slot past the row's last one
    };
    /// Work in the row of subsequence `sub` of the scan, which is subsequence `rel` of its segment.
    __device__ __forceinline__ void set_row(int sub, int rel)
    {
        const uint32_t u = static_cast<uint32_t>(sub);
        row0             = (u >> kLog2Rows) * kRowBytes + (u & ((1u << kLog2Rows) - 1u)) * 4u;
        row_word0        = rel * W;
    }
    __device__ __forceinline__ Pos start(int w) const
    {
        return Pos{row0 + static_cast<uint32_t>(w - row_word0 + kRowLeadWords) * kSlotBytes, row0 + kRowBytes};
    }
    __device__ __forceinline__ void advance(Pos& q) const
    {
        q.off += kSlotBytes;
        if (kCrossRows && __builtin_expect(q.off == q.end, 0)) { // behind slot W + 2: the stream goes on at slot 3 of the next row
            // a real branch: as selects these eight instructions run with every refill of every wave, while a lane
            // leaves its row once or twice per subsequence (the empty statement keeps the compiler from if-converting)
            asm volatile("");
            const uint32_t step = (q.off & (kSlotBytes - 1u)) == kSlotBytes - 4u ? kRowBytes - (kSlotBytes - 4u) : 4u; // last row of a tile: on to the next tile
            q.off               = q.off - W * kSlotBytes + step;
            q.end += step;
        }
    }
    __device__ __forceinline__ uint32_t load(const Pos& q) const
    {
#if defined(JG_EXP_FAKE_REFILL) // timing experiments only: no memory operation in the symbol loops (results are garbage)
        if (kCrossRows) return q.off * 2654435761u ^ (q.off >> 7);
#endif
        return *reinterpret_cast<JG_GLOBAL const uint32_t*>(reinterpret_cast<JG_GLOBAL const uint8_t*>(scan32) + q.off);
    }
    __device__ __forceinline__ uint32_t cook(uint32_t v, const Pos&) const { return v; }
};

/// The write pass's view of the bitstream (jg_huff_core.h, decode_units): a 64-bit window `hi`:`lo` looked at `sh`
/// bits from the right (BitWindow's funnel shift), `nxt` the word behind it, in flight, and `pos` the byte offset of
/// that word in the tiled destuffed buffer (GlobalFetch<W, true>: the same rows and addresses; behind the row's last
/// slot, W + 2, the stream goes on at slot 3 of the next row).
///
/// What is special is the refill. Loaded the obvious way -- `if (sh < 0) { hi = lo; lo = nxt; nxt = load(next); }` --
/// every iteration of the loop waits for the word loaded one iteration earlier: some lane of the wave refills in
/// nearly every iteration, the wait is the wave's, it is a wait for ALL its loads (s_waitcnt vmcnt(0): the compiler
/// cannot tell them apart), and under this kernel's load a load takes longer than an iteration: a quarter of the
/// kernel's time (measured with the loads compiled out). The lane that refills now, though, needs the word it asked for
/// at its PREVIOUS refill, a handful of iterations ago. So the loads are issued from inline assembly, where the compiler
/// does not see them, and the wait is for all but the most recent one (vmcnt(1)). That is enough because
///   * vector memory operations of a wave complete in the order they were issued: `s_waitcnt vmcnt(N)` waits until all
///     but the wave's N youngest vector-memory operations are done, and loads, stores and atomics of the global_* / buffer_*
///     forms share that ONE counter in issue order (only flat_* operations may return out of order, and this kernel has
///     none: every pointer of the job is qualified address_space(1), JobView) -- /opt/skills/guides/MI355X_MICROARCH.md,
///     "s_waitcnt vmcnt(N)", and the gfx9-family ISA manuals' description of VM_CNT. The sink's stores, which the
///     COMPILER issues between two refills, are operations on the same counter like any other: one issued behind a
///     refill's load makes that load older, so vmcnt(1) waits for it as well (stricter than needed, never laxer); one
///     issued in front of it is simply waited for. What vmcnt(1) never waits for is the wave's YOUNGEST operation, which
///     is why the word a lane consumes must come from a load that is not the youngest:
///   * a lane does not refill in two consecutive iterations (`ok`: a lane that would sits one iteration out; it would
///     need 33 bits in two symbols), and
///   * a lane that refilled one iteration ago loads the same word into `nxt` AGAIN (`prev`): behind the load a lane
///     needs there is therefore always a younger one -- its own --, whatever the other lanes do.
/// Everything that touches `nxt` lives in that assembly block, with read-write operands: the compiler never copies the
/// register while a load into it is in flight -- which it does not KNOW, so the build checks it (jpeggpu_amd/build.py,
/// check_refill: no scratch and no spills in huff_write, and between the loop's first refill and done() no instruction
/// outside these assembly blocks names the register; otherwise the library is rebuilt with -DJG_SAFE_REFILL: vmcnt(0)). Operations the compiler does track (the sink's stores) only make its
/// own waits stricter. The refill is branch-free: an exec mask around three moves, two adds and the load.
template <int W>
struct RowWindow {
    typedef GlobalFetch<W, true> G;
    G g;
    uint32_t hi, lo, nxt;
    uint32_t pos, last; // byte offset of the word in `nxt`; offset of the row's last slot
    int sh;
    uint32_t ok;        // ~0 where the lane may refill in this iteration, 0 where it did in the previous one
    __device__ __forceinline__ void seek(int p)
    {
        const int q             = p - 1; // the pair starts at the word that holds bit p - 1: sh stays in 0..31
        sh                      = 31 - (q & 31);
        typename G::Pos at      = g.start(q >> 5); // arithmetic shift: -1 for q == -1
        hi                      = g.load(at);
        g.advance(at);
        lo = g.load(at);
        g.advance(at);
        nxt  = g.load(at);
        pos  = at.off;
        last = at.end - G::kSlotBytes;
        ok   = ~0u;
        // the three words must have ARRIVED before the loop: its own waits count loads, and these are not among them
        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2" : "+v"(hi), "+v"(lo), "+v"(nxt));
    }
    __device__ __forceinline__ void top()
    {
        const uint32_t t              = static_cast<uint32_t>(sh) & ok;
        const unsigned long long need = __builtin_amdgcn_ballot_w64(static_cast<int>(t) < 0);
        const unsigned long long both = __builtin_amdgcn_ballot_w64(static_cast<int>(t | ~ok) < 0); // + the lanes that refilled one iteration ago
        unsigned long long save;
        ok = ~0u;
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "s_and_b64 exec, %[save], %[need]\n\t"
#if defined(JG_SAFE_REFILL) // the build falls back to this when its check of the generated code fails (jpeggpu_amd/build.py)
            "s_waitcnt vmcnt(0)\n\t"
#else
            "s_waitcnt vmcnt(1)\n\t"
#endif
            "v_mov_b32 %[hi], %[lo]\n\t"
            "v_mov_b32 %[lo], %[nxt]\n\t"
            "v_add_u32 %[pos], %[step], %[pos]\n\t"
            "v_add_u32 %[sh], 32, %[sh]\n\t"
            "v_mov_b32 %[ok], 0\n\t"
            "s_and_b64 exec, %[save], %[both]\n\t"
            "global_load_dword %[nxt], %[pos], %[base]\n\t"
            "s_mov_b64 exec, %[save]"
            : [hi] "+v"(hi), [lo] "+v"(lo), [nxt] "+v"(nxt), [pos] "+v"(pos), [sh] "+v"(sh), [ok] "+v"(ok), [save] "=&s"(save)
            : [need] "s"(need), [both] "s"(both), [step] "s"(G::kSlotBytes), [base] "s"(g.scan32)
            : "memory", "scc");
    }
    /// Behind the loop: the last loads into `nxt` may still be in flight, and the compiler, which does not know of them,
    /// is free to give the register to something else from here on.
    __device__ __forceinline__ void done() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt) : : "memory"); }
    __device__ __forceinline__ uint32_t look() const { return __builtin_amdgcn_alignbit(hi, lo, static_cast<uint32_t>(sh)); }
    __device__ __forceinline__ void skip(int n) { sh -= n; }
    __device__ __forceinline__ int left() const { return sh; }
    /// Negative where `nxt` is the word of the row's last slot: the next refill must not step to the slot behind it.
    __device__ __forceinline__ int crossed() const { return static_cast<int>((pos ^ last) - 1u); } // offsets are below 2^31
    __device__ __forceinline__ void cross()
    {
        // `pos` is the row's last slot, W + 2, whose word is also slot 2 of the next row (jg_defs.h: the mirrored words):
        // go there, the next refill then steps to slot 3, where the stream goes on (GlobalFetch::advance). A lane that
        // loads `nxt` again after this (top(), `prev`) gets the same word from the new place.
        const uint32_t step = (pos & (G::kSlotBytes - 1u)) == G::kSlotBytes - 4u ? G::kRowBytes - (G::kSlotBytes - 4u) : 4u; // last row of a tile: on to the next tile
        pos                 = pos - W * G::kSlotBytes + step;
        last += step;
    }
};

#if defined(JG_PROBE)
// Probe builds only (python jpeggpu_amd/build.py out.so -DJG_PROBE): 100 MHz time stamps of the passes of
// huff_sync_intra, 64 per workgroup; read back with jpeggpu_probe_read (tools/probe/sync_stamps.py).
__device__ uint32_t g_probe[4096 * 64];
// huff_write: [0] sum over lanes of loop iterations (low 32 bits), [1] high bits, [2] largest, [3] lanes (tools/probe/write_iters.py)
__device__ unsigned long long g_probe_write[4];
// huff_write, the launch's first job, by subsequence: loop iterations of the lane's wave, then (from 1 << 16) symbols the
// lane decoded (tools/probe/write_lane_iters.py)
__device__ uint16_t g_probe_lane_iters[1 << 17];
__device__ uint32_t g_probe_lane_rare[1 << 15]; // times the lane's wave took the rare block | times the lane asked for it << 16
// huff_sync_tail, jobs 0..15 x parts 0..7 of a launch: 100 MHz stamps [0] start, [1] flow list built, [2 + k] after
// trip k of the flow loop (all groups), [63] = flows | trips << 16 (tools/probe/tail_stamps.py)
__device__ uint32_t g_probe_tail[128 * 64];
#define JG_TAIL_STAMP(i, v)                                                                               \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.y < 16 && blockIdx.x < 8 && (i) < 64) g_probe_tail[(blockIdx.y * 8 + blockIdx.x) * 64 + (i)] = (v); \
    } while (0)
#define JG_STAMP(i)                                                                                       \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 4096 && (i) < 64) g_probe[blockIdx.x * 64 + (i)] = static_cast<uint32_t>(wall_clock64()); \
    } while (0)
#else
#define JG_STAMP(i) do { } while (0)
#define JG_TAIL_STAMP(i, v) do { } while (0)
#endif

/// A pointer every lane of the wave holds the same value of, moved to scalar registers where the compiler cannot see that
/// (huff_tail_write, whose job follows from a ticket): free where it can.
template <class P>
__device__ __forceinline__ P uniform_ptr(P p)
{
    const uint64_t v  = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<P>(static_cast<uint64_t>(hi) << 32 | lo);
}

/// Loads and stores of the words workgroups of ONE launch hand to each other (huff_tail_write: the states and sequence
/// tails its parts finish, which its sequences read): kCoherent makes them agent-scope relaxed atomics, which on gfx950 are
/// plain loads and stores that go to the level of the memory hierarchy all eight XCDs share (sc1) -- every XCD has an L2 of
/// its own. The alternative, ordinary accesses between a release fence in the writer and an acquire fence in the reader,
/// costs a write-back of the XCD's whole L2 per writing wave and an invalidation of it per reading wave: measured, the
/// fused launch took 1 085 us where the two kernels it replaces take 931.
template <bool kCoherent, class T_>
__device__ __forceinline__ void st_shared(JG_GLOBAL T_* p, T_ v)
{
    if constexpr (kCoherent) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool kCoherent, class T_>
__device__ __forceinline__ T_ ld_shared(JG_GLOBAL const T_* p)
{
    if constexpr (kCoherent) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

/// LDS address of a pointer into the workgroup's shared memory.
__device__ __forceinline__ uint32_t lds_address(const void* p)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*)p));
}

/// Copy the scan's Huffman table pack (a multiple of 16 bytes) into LDS and make the offsets of the cursor ring
/// at its end (from `sp.cursor_off` on: table offsets in the two halves of word 0, own and next entry in words 2
/// and 3) absolute LDS addresses, as is `sp.cursor_off` afterwards (jg_huff_core.h, JG_TAB_AT).
__device__ __forceinline__ void load_tables(uint8_t* s_tab, JG_GLOBAL const uint8_t* __restrict__ g_tab, ScanParams& sp)
{
    uint4* d       = reinterpret_cast<uint4*>(s_tab);
    JG_GLOBAL const uint4* s = reinterpret_cast<JG_GLOBAL const uint4*>(g_tab);
    const uint32_t base = lds_address(s_tab), ring = sp.cursor_off / 16;
    // Four loads of a lane in flight at a time, no branch (a piece past the end is the last one again): one
    // at a time, every 4 KB of the pack was a memory latency of its own in front of the workgroup's first symbol --
    // nine in a row for a 36 KB sync pack.
    const uint32_t n16 = sp.tab_bytes / 16, step = blockDim.x;
    for (uint32_t i0 = threadIdx.x; i0 < n16; i0 += 4 * step) {
        uint4 v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) v[k] = ld_global(s + min(i0 + k * step, n16 - 1));
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i   = min(i0 + k * step, n16 - 1); // (the last piece written again with the same content)
            const uint32_t add = i >= ring ? base : 0u;
            v[k].x += add * 0x00010001u; // LDS addresses stay below 64 KiB: static_asserts at the carves, launch_huff at run time
            v[k].z += add;
            v[k].w += add;
            d[i] = v[k];
        }
    }
    sp.cursor_off += base;
}

/// Carve of the dynamic LDS of the two sequence-wide Huffman kernels.
struct SeqLds {
    static constexpr uint32_t kState = 0;                          // 5 state arrays of T + 1 words
    static constexpr uint32_t kFlows = kState + 5 * (T + 1) * 4;   // 4 arrays of T words: the flows re-packed between iterations
    static constexpr uint32_t kPend  = kFlows + 4 * T * 4;         // T + 1 bytes of marks
    static constexpr uint32_t kTabs  = (kPend + T + 1 + 15) / 16 * 16;
    static_assert(kTabs % 16 == 0, "the table pack is read with 16-byte loads");
};
/// Static LDS of a kernel precedes its dynamic LDS; the Huffman kernels declare at most this much.
constexpr uint32_t kStaticLdsSlack = 256;
static_assert(kStaticLdsSlack + SeqLds::kTabs + kMaxTablePackSync <= 65536 && kStaticLdsSlack + 6 * (T + 1) * 4 + 80 + kMaxTablePackSync <= 65536, "absolute table addresses are packed into 16 bits (load_tables)");

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_add_u16(a, b); }

/// Inclusive block-wide position of the set flags among TL lanes; returns the block total.
template <int TL>
__device__ __forceinline__ int block_rank(bool flag, int* s_wave, int& rank)
{
    const unsigned long long m = __ballot(flag);
    const int before           = __popcll(m & ((1ull << lane_id()) - 1ull));
    const int w                = threadIdx.x >> 6;
    __syncthreads(); // previous use of s_wave is over
    if (lane_id() == 0) s_wave[w] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < TL / 64; ++k) {
        const int c = s_wave[k];
        off += k < w ? c : 0;
        total += c;
    }
    rank = off + before;
    return total;
}

// ------------------------------------------------------------------------------------------------
// Huffman: speculative decode + intra-sequence synchronisation
// ------------------------------------------------------------------------------------------------

/// Lane t decodes subsequence first_sub - OV + t from the guessed state (c, z) = (0, 0) (exit state
/// only), then decodes the following subsequences of the same segment -- now with n and the DC sums --
/// until the state it reaches equals the one stored there. A subsequence that
/// opens a segment is decoded from the segment's start state by its left neighbour. In iteration i
/// entry j = t+1+i of the LDS state table is read and written by the flow that started at lane t only, so one
/// workgroup barrier per iteration is enough. At least one iteration is needed (max_intra_iters >= 1).
///
/// FLOWS ARE RE-PACKED. The first flow iteration is every lane's (it supplies n and the DC sums of every
/// subsequence). After it a few lanes in a hundred still flow -- 8 % on the 12 MP synthetic images, 22 % on the
/// reference's photo at 256-byte subsequences (tools/probe/flow_study.py), and nearly all of them only CONFIRM the
/// next entry: 99.5 % / 94 % of the table is already the sequential decoder's. Left where they are they keep all four
/// waves of the workgroup in the loop; handed to huff_sync_tail (round 3) they become a chain of whole-subsequence
/// decodes on a chip with one wave per SIMD, 430 us per 64 images of pure latency. So the survivors are packed into
/// the lowest lanes between iterations (state through LDS: the entry a flow continues from may be overwritten in the
/// same iteration by the flow behind it): the follow-up iterations cost ONE wave per workgroup, a workgroup leaves as
/// soon as it has no flow left, and what the tail kernel still gets is the rare flow that crosses into the next
/// sequence. This is the kernel of LONE decodes (one image: every flow stays in its sequence's workgroup). For batches
/// it was measured and lost -- the workgroup keeps its 22 KB of tables while one of its four waves works, and LDS is what
/// bounds this kernel when the chip is full: huff_sync_intra_batch below.
///
/// The first OV lanes re-decode the last OV subsequences of the PREVIOUS sequence (their results are
/// not stored): their flows enter this sequence the way the previous workgroup's would, so the first
/// subsequences of the sequence are normally already what the inter-sequence kernel will confirm. The exit state of
/// the previous sequence's last subsequence AS THIS WORKGROUP SAW IT goes to `bnd_p / bnd_cz`: where it equals what the
/// previous workgroup stored the boundary needs no flow at all (huff_sync_tail).
template <int W, class JS>
__global__ __launch_bounds__(T) void huff_sync_intra(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_wave[T / 64];
    __shared__ int s_cut; // a flow was cut short right at the last entry of the overlap zone: bnd would not be what entry OV came from
    typedef SeqLds Lds;
    int* s_p         = reinterpret_cast<int*>(smem + Lds::kState);
    int* s_n         = s_p + (T + 1);
    int* s_cz        = s_n + (T + 1);
    uint32_t* s_dc01 = reinterpret_cast<uint32_t*>(s_cz + (T + 1));
    uint32_t* s_dc23 = s_dc01 + (T + 1);
    int* s_fj        = reinterpret_cast<int*>(smem + Lds::kFlows); // re-packed flows: entry last written | lane limit << 16,
    int* s_frel      = s_fj + T;                                      //   that entry's index in its segment,
    int* s_fp        = s_frel + T;                                    //   the state reached there
    int* s_fcz       = s_fp + T;
    uint8_t* s_pend  = smem + Lds::kPend; // bit 0: a flow that wrote the entry is unfinished; bit 1: a state a flow may stop at
    uint8_t* s_tab   = smem + Lds::kTabs;

    const JobView J(js.get());
    if (static_cast<int>(blockIdx.x) >= J.num_seq) return;
    ScanParams sp = J.sp;
    sp.use_sync_pack();
    const int t         = threadIdx.x;
    s_pend[t]           = 0;
    if (t == 0) s_cut = 0;
    const int SEQ = sp.seq_subseq, OV = T - SEQ;             // 240 + 16 for a lone decode, 255 + 1 in batches (jg_defs.h)
    const int first_sub = blockIdx.x * SEQ;                  // first subsequence this workgroup owns
    const int img_first = first_sub - OV;                    // subsequence of lane 0
    const int img_end   = min(T, sp.num_subseq - img_first); // lanes below this have a subsequence

    JG_STAMP(0);
    load_tables(s_tab, J.tables_sync, sp);
    __syncthreads();
    JG_STAMP(1);

    constexpr int kBits = W * 32;
    const int sub       = img_first + t;
    const bool active   = sub >= 0 && t < img_end;
    // Multi-hypothesis speculation (jg_defs.h): the table this kernel starts from was written by huff_mh_resolve -- for
    // every subsequence the candidate state the chain of links passes through -- instead of being speculated here. The
    // flows below then are the reference's, from that table; an entry the chain hopped over (not `known`) starts no
    // flow and stops none: the flow from upstream fills it in.
    const bool mh = sp.mh > 1; // set for decodes that ran the three kernels in front of this one (lone decodes)
    LaneState st{};
    BitWindow<GlobalFetch<W>> bw{};
    GlobalFetch<W> fetch{reinterpret_cast<JG_GLOBAL const uint32_t*>(J.destuffed), 0, 0};
    Segment seg{0, 0};
    int rel = 0;
    bool known = true;
    if (active) {
        seg = ld_global(J.segments + J.seg_idx[sub]);
        rel = sub - seg.subseq_offset;
        if (mh) {
            // (the resolved table has its own buffers, hypothesis 0's: what another workgroup of THIS launch stores into
            // st_p / st_cz is never read here)
            st.p         = J.mh_p[sub];
            const int cz = J.mh_cz[sub];
            st.c         = cz & 0xFF;
            st.z         = cz >> 8;
            known        = J.mh_known[sub] != 0;
            s_pend[t]    = known ? 2 : 0;
        } else {
            // speculative pass: the own subsequence from the guessed state (c, z) = (0, 0), exit state only
            st.p = rel * kBits;
            fetch.set_row(sub, rel);
            bw.seek(st.p, fetch);
            if (JS::kSpeculateStateOnly) {
                SpecSink none;
                decode_subsequence(st, bw, fetch, (rel + 1) * kBits, s_tab, sp, none);
            } else {
                NoSink sums; // exact for a subsequence that opens a segment, replaced by a flow everywhere else
                decode_subsequence(st, bw, fetch, (rel + 1) * kBits, s_tab, sp, sums);
            }
            if (!JS::kSpeculateStateOnly) {
                s_n[t]    = st.n;
                s_dc01[t] = st.dc01;
                s_dc23[t] = st.dc23;
            }
            s_pend[t] = 2;
        }
        s_p[t]  = st.p;
        s_cz[t] = st.c | (st.z << 8);
    }
    __syncthreads();
    JG_STAMP(2);

    // Flow passes. Lane t first decodes subsequence j = t + 1 from the own exit state. With a state-only
    // speculative pass (and with the multi-hypothesis table) it also does so if j opens a restart segment -- from the
    // segment's start state, which is known, not guessed -- so that every subsequence gets its n and DC sums from a
    // decode that started in a real state; otherwise the speculative pass of such a j was already exact.
    // (huff_sync_tail relies on this: whichever way, a segment's first subsequence ends up decoded from the segment's start
    // state -- here by the flow of the lane in front of it, else by its own speculative pass, which started there)
    const bool covers_starts = JS::kSpeculateStateOnly || mh;
    bool flowing = covers_starts ? t + 1 < img_end && img_first + t + 1 >= 0 : active;
    int lim      = 0; // flows stay below this lane index: end of the segment or of the image
    if (flowing) {
        const int sub_j = img_first + t + 1;
        if (covers_starts && (!active || rel + 1 == seg.subseq_count)) { // j opens the next segment
            seg   = ld_global(J.segments + J.seg_idx[sub_j]);
            rel   = -1;
            st    = LaneState{};
            known = true;
        }
        lim = min(img_end, seg.subseq_offset + seg.subseq_count - img_first);
        flowing = known;
    }
    NoSink sink;
    // One step of a flow that has reached entry j - 1 with state `st` (`rel`: that entry's index in its segment):
    // decode entry j, compare, store. Returns whether the flow goes on.
    const auto flow_step = [&](int j) -> bool {
        st.n    = 0;
        st.dc01 = 0;
        st.dc23 = 0;
        // every decode works in the row of its subsequence (GlobalFetch): the window is set up again from p
        ++rel;
        fetch.set_row(img_first + j, rel);
        bw.seek(st.p, fetch);
        decode_subsequence(st, bw, fetch, (rel + 1) * kBits, s_tab, sp, sink);
        const int cz     = st.c | (st.z << 8);
        const bool synced = st.p == s_p[j] && cz == s_cz[j] && (s_pend[j] & 2); // still store n / dc
        s_p[j]    = st.p;
        s_n[j]    = st.n;
        s_cz[j]   = cz;
        s_dc01[j] = st.dc01;
        s_dc23[j] = st.dc23;
        s_pend[j] = 2; // from here on a state a flow may stop at
        return !synced;
    };
    // One loop, one copy of the symbol loop in it. The first iteration is every lane's own flow; `j` is the entry the
    // lane's flow wrote last.
    int j    = t;
    int iter = 0;
    for (; iter < sp.max_intra_iters; ++iter) {
        bool go = flowing && j + 1 < lim;
        if (iter > 0) {
            // survivors into the lowest lanes (block_rank's barriers also order this iteration's table accesses behind
            // the previous one's)
            int rank;
            const int total = block_rank<T>(go, s_wave, rank);
            if (total == 0) break;
            if (go) {
                s_fj[rank]   = j | lim << 16;
                s_frel[rank] = rel;
                s_fp[rank]   = st.p;
                s_fcz[rank]  = st.c | (st.z << 8);
            }
            __syncthreads();
            go = t < total;
            if (go) {
                const int pk = s_fj[t], cz = s_fcz[t];
                j    = pk & 0xFFFF;
                lim  = pk >> 16;
                rel  = s_frel[t];
                st.p = s_fp[t];
                st.c = cz & 0xFF;
                st.z = cz >> 8;
            }
        }
        flowing = go;
        if (go) flowing = flow_step(++j);
        JG_STAMP(3 + iter);
    }
    // Flows cut short by the iteration cap continue in huff_sync_tail from the entry they reached
    // last: mark that entry (flows still inside the overlap zone belong to the previous workgroup).
    if (iter == sp.max_intra_iters && flowing && j + 1 < lim) {
        if (j >= OV) s_pend[j] |= 1;
        if (j == OV - 1) s_cut = 1;
    }
    __syncthreads();

    if (active && t >= OV) {
        J.pending[sub] = static_cast<uint8_t>(s_pend[t] & 1);
        J.st_p[sub]    = s_p[t];
        J.st_n[sub]    = s_n[t];
        J.st_cz[sub]   = s_cz[t];
        J.st_dc01[sub] = s_dc01[t];
        J.st_dc23[sub] = s_dc23[t];
    }
    if (t == OV - 1) {
        // what entry OV (the sequence's first subsequence) was decoded from, if anything in this workgroup says so
        const bool have = active && !s_cut;
        J.bnd_p[blockIdx.x]  = have ? s_p[t] : -1;
        J.bnd_cz[blockIdx.x] = have ? s_cz[t] : -1;
    }
}

/// The sequence kernel of a BATCH (JobArray): round 3's kernel as it stood -- every flow stays in its lane, the loop is
/// capped (normally at its first iteration) and the rest goes to huff_sync_tail. The kernel above computes the same with
/// max_intra_iters == 1 and takes the same time alone on the chip (536 against 538 us per 64 images), but a batch whose
/// four streams overlap ran 5 % slower with it (26.5 k against 27.9 k images/s, measured six times in-run against this
/// one; neither its LDS footprint -- padded to this kernel's and beyond --, nor its code size, nor the job's layout, nor
/// the boundary records explain it: DESIGN.md section 7), so batches keep this one.
/// Sequences a workgroup of the batch kernel takes, each on T of its lanes, all behind ONE copy of the table pack: what
/// a CU holds of this kernel is bound by LDS, most of it the pack, and the kernel's time by how many waves a SIMD has to
/// choose from (5 -> 4 workgroups per CU cost 21 %, round 4): two sequences per pack are 8 waves per SIMD instead of 5.
#ifndef JG_BATCH_SEQ_PER_WG
#define JG_BATCH_SEQ_PER_WG 2
#endif
constexpr int kBatchSeqPerWg = JG_BATCH_SEQ_PER_WG;
struct SeqLdsBatch {
    static constexpr uint32_t kState = 0;
    static constexpr uint32_t kOne   = (6 * (T + 1) * 4 + 64 + 15) / 16 * 16; // one sequence's 5 state arrays / scan scratch
    static constexpr uint32_t kTabs  = kState + kBatchSeqPerWg * kOne;
    static_assert(kTabs % 16 == 0, "the table pack is read with 16-byte loads");
};
static_assert(kStaticLdsSlack + SeqLdsBatch::kTabs + kMaxTablePackSync <= 65536, "absolute table addresses are packed into 16 bits (load_tables)");
/// Control words of huff_tail_write (below), ScanJob::fuse_ctl of every job of the launch, set up by fuse_init from the
/// sequence kernel of the same call:
///   [0]                      the launch's ticket counter                              (the FIRST job's words are used)
///   [kFuseJobs + 2 j + 0/1]  job j's ready queue: entries pushed / entries claimed     (the first job's, j < kFuseMaxJobs)
///   [kFuseOwn + s]           parts of the tail pass that hold subsequences of sequence s and are not done yet
///   [kFuseOwn + num_seq + i] the job's ready queue: i-th sequence whose parts are all done (kFuseEmpty: not there yet)
constexpr int kFuseJobs = 16, kFuseMaxJobs = 256, kFuseOwn = kFuseJobs + 2 * kFuseMaxJobs;
constexpr uint32_t kFuseEmpty = 0xFFFFFFFFu;
static_assert(fuse_ctl_words(0) == static_cast<size_t>(kFuseOwn), "the plan carves what fuse_init lays out (jg_defs.h)");

__device__ __forceinline__ void fuse_init(const ScanJob* jobs, int job, const JobView& J, int lanes)
{
    JG_GLOBAL uint32_t* own  = as_global(jobs[job].fuse_ctl);
    JG_GLOBAL uint32_t* ctl0 = as_global(jobs[0].fuse_ctl);
    const int t = threadIdx.x, SEQ = J.sp.seq_subseq, S = J.sp.num_subseq;
    if (t < kFuseJobs) own[t] = 0u;
    if (t < 2 && job < kFuseMaxJobs) ctl0[kFuseJobs + 2 * job + t] = 0u;
    for (int q = t; q < J.num_seq; q += lanes) {
        const int first = q * SEQ, last = min(first + SEQ, S) - 1;
        uint32_t holders = 0;
        for (int k = 0; k < J.num_tail_parts; ++k) holders += J.tail_parts[k] <= last && J.tail_parts[k + 1] > first ? 1u : 0u;
        own[kFuseOwn + q]             = holders;
        own[kFuseOwn + J.num_seq + q] = kFuseEmpty;
    }
}

template <int W, class JS>
__global__ __launch_bounds__(T * kBatchSeqPerWg) void huff_sync_intra_batch(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int which = threadIdx.x / T;                       // the workgroup's sequence this lane works for
    const int seq   = blockIdx.x * kBatchSeqPerWg + which;
    int* s_p         = reinterpret_cast<int*>(smem + SeqLdsBatch::kState + which * SeqLdsBatch::kOne);
    int* s_n         = s_p + T;
    int* s_cz        = s_n + T;
    uint32_t* s_dc01 = reinterpret_cast<uint32_t*>(s_cz + T);
    uint32_t* s_dc23 = s_dc01 + T;
    int* s_pend      = reinterpret_cast<int*>(s_dc23 + T);
    uint8_t* s_tab   = smem + SeqLdsBatch::kTabs;

    const JobView J(js.get());
    if constexpr (std::is_same<JS, JobArray>::value) {
        if (blockIdx.x == 0) fuse_init(js.jobs, static_cast<int>(blockIdx.y), J, T * kBatchSeqPerWg); // huff_tail_write of the same call
    }
    if (static_cast<int>(blockIdx.x) * kBatchSeqPerWg >= J.num_seq) return;
    ScanParams sp = J.sp;
    sp.use_sync_pack();
    const int t         = threadIdx.x % T;
    s_pend[t]           = 0;
    const int SEQ = sp.seq_subseq, OV = T - SEQ;             // 240 + 16 for a lone decode, 255 + 1 in batches (jg_defs.h)
    // (a sequence behind the scan's last one has no lane with a subsequence: its lanes only keep the barriers company)
    const int first_sub = seq * SEQ;                         // first subsequence this sequence owns
    const int img_first = first_sub - OV;                    // subsequence of lane 0
    const int img_end   = min(T, sp.num_subseq - img_first); // lanes below this have a subsequence

    constexpr int kBits = W * 32;
    const int sub       = img_first + t;
    const bool active   = sub >= 0 && t < img_end;
    // (the lane's segment: its index asked for before the table pack is copied, the record right behind it -- two memory
    // latencies that used to follow the copy)
    const int lane_seg = J.seg_idx[active ? sub : 0];

    JG_STAMP(0);
    load_tables(s_tab, J.tables_sync, sp);
    const Segment lane_segment = ld_global(J.segments + lane_seg);
    __syncthreads();
    JG_STAMP(1);

    // Multi-hypothesis speculation (jg_defs.h): the table this kernel starts from was written by huff_mh_resolve -- for
    // every subsequence the candidate state the chain of links passes through -- instead of being speculated here. The
    // flows below then are the reference's, from that table; an entry the chain hopped over (not `known`) starts no
    // flow and stops none: the flow from upstream fills it in.
    // (a batch never runs the multi-hypothesis kernels, build_jobs: the branches below fold away -- and must: with the
    // table read from mh_p / mh_cz in the dead branch, two more pointers of the job loaded at the kernel's start, a batch
    // of four overlapping streams ran 5 % slower, measured in-run eight times over)
    constexpr bool mh = false;
    LaneState st{};
    BitWindow<GlobalFetch<W>> bw{};
    GlobalFetch<W> fetch{reinterpret_cast<JG_GLOBAL const uint32_t*>(J.destuffed), 0, 0};
    Segment seg{0, 0};
    int rel = 0;
    bool known = true;
    if (active) {
        seg = lane_segment;
        rel = sub - seg.subseq_offset;
        if (mh) {
            st.p         = J.mh_p[sub];
            const int cz = J.mh_cz[sub];
            st.c         = cz & 0xFF;
            st.z         = cz >> 8;
            known        = J.mh_known[sub] != 0;
            s_pend[t]    = known ? 2 : 0; // bit 1: the entry is a state some flow may stop at
        } else {
            // speculative pass: the own subsequence from the guessed state (c, z) = (0, 0), exit state only
            st.p = rel * kBits;
            fetch.set_row(sub, rel);
            bw.seek(st.p, fetch);
            if (JS::kSpeculateStateOnly) {
                SpecSink none;
                decode_subsequence(st, bw, fetch, (rel + 1) * kBits, s_tab, sp, none);
            } else {
                NoSink sums; // exact for a subsequence that opens a segment, replaced by a flow everywhere else
                decode_subsequence(st, bw, fetch, (rel + 1) * kBits, s_tab, sp, sums);
                s_n[t]    = st.n;
                s_dc01[t] = st.dc01;
                s_dc23[t] = st.dc23;
            }
            s_pend[t] = 2;
        }
        s_p[t]  = st.p;
        s_cz[t] = st.c | (st.z << 8);
    }
    __syncthreads();
    JG_STAMP(2);

    // Flow passes. Lane t first decodes subsequence j = t + 1 from the own exit state. With a state-only
    // speculative pass (and with the multi-hypothesis table) it also does so if j opens a restart segment -- from the
    // segment's start state, which is known, not guessed -- so that every subsequence gets its n and DC sums from a
    // decode that started in a real state; otherwise the speculative pass of such a j was already exact.
    const bool covers_starts = JS::kSpeculateStateOnly || mh;
    bool flowing = covers_starts ? t + 1 < img_end && img_first + t + 1 >= 0 : active;
    int lim      = 0; // flows stay below this lane index: end of the segment or of the image
    if (flowing) {
        const int sub_j = img_first + t + 1;
        if (covers_starts && (!active || rel + 1 == seg.subseq_count)) { // j opens the next segment
            seg   = ld_global(J.segments + J.seg_idx[sub_j]);
            rel   = -1;
            st    = LaneState{};
            known = true;
        }
        lim = min(img_end, seg.subseq_offset + seg.subseq_count - img_first);
        flowing = known;
    }
    NoSink sink;
    int iter = 0;
    for (; iter < sp.max_intra_iters; ++iter) {
        const int j = t + 1 + iter;
        if (flowing && j < lim) {
            st.n    = 0;
            st.dc01 = 0;
            st.dc23 = 0;
            // every decode works in the row of its subsequence (GlobalFetch): the window is set up again from p
            ++rel;
            fetch.set_row(img_first + j, rel);
            bw.seek(st.p, fetch);
            decode_subsequence(st, bw, fetch, (rel + 1) * kBits, s_tab, sp, sink);
            const int cz = st.c | (st.z << 8);
            if (st.p == s_p[j] && cz == s_cz[j] && (s_pend[j] & 2)) flowing = false; // synchronised; still store n / dc
            s_p[j]    = st.p;
            s_n[j]    = st.n;
            s_cz[j]   = cz;
            s_dc01[j] = st.dc01;
            s_dc23[j] = st.dc23;
            s_pend[j] = 2; // from here on a state a flow may stop at
        } else {
            flowing = false;
        }
        const bool more = __syncthreads_or(flowing && j + 1 < lim);
        JG_STAMP(3 + iter);
        if (!more) break;
    }
    // Flows cut short by the iteration cap continue in huff_sync_tail from the entry they reached
    // last: mark that entry (flows still inside the overlap zone belong to the previous workgroup).
    if (iter == sp.max_intra_iters && flowing && t + 1 + iter < lim && t + iter >= OV) s_pend[t + iter] |= 1;
    __syncthreads();

    if (active && t >= OV) {
        J.pending[sub] = static_cast<uint8_t>(s_pend[t] & 1);
        J.st_p[sub]    = s_p[t];
        J.st_n[sub]    = s_n[t];
        J.st_cz[sub]   = s_cz[t];
        J.st_dc01[sub] = s_dc01[t];
        J.st_dc23[sub] = s_dc23[t];
    }
    // (this kernel says nothing about the boundary: huff_sync_tail then starts a flow at every one, as round 3 did)
    if (t == OV - 1 && seq < J.num_seq) {
        J.bnd_p[seq]  = -1;
        J.bnd_cz[seq] = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: multi-hypothesis speculation (one image at a time; jg_defs.h)
// ------------------------------------------------------------------------------------------------

/// Candidate exit states: lane (sub, h = blockIdx.y) decodes subsequence `sub` as if a data unit of index h started at
/// its first bit (hypothesis 0 is the reference's speculation, decode_huffman.cu:457-467).
template <int W, class JS>
__global__ __launch_bounds__(T) void huff_mh_spec(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const JobView J(js.get());
    ScanParams sp = J.sp;
    sp.use_sync_pack();
    load_tables(smem, J.tables_sync, sp);
    __syncthreads();
    const int sub = blockIdx.x * T + threadIdx.x, h = blockIdx.y;
    if (sub == 0 && h == 0) st_global(J.mh_pool, uint2_t{0u, 0u}); // nothing handed out yet (huff_mh_flow counts)
    if (sub >= sp.num_subseq) return;
    const Segment seg = ld_global(J.segments + J.seg_idx[sub]);
    const int rel     = sub - seg.subseq_offset;
    LaneState st{};
    st.p = rel * (W * 32);
    st.c = h;
    GlobalFetch<W> fetch{reinterpret_cast<JG_GLOBAL const uint32_t*>(J.destuffed), 0, 0};
    fetch.set_row(sub, rel);
    BitWindow<GlobalFetch<W>> bw{};
    bw.seek(st.p, fetch);
    SpecSink none;
    decode_subsequence(st, bw, fetch, (rel + 1) * (W * 32), smem, sp, none);
    const size_t at = static_cast<size_t>(h) * sp.num_subseq + sub;
    J.mh_p[at]      = st.p;
    J.mh_cz[at]     = st.c | (st.z << 8);
}

/// Links: lane (sub, h) decodes the subsequences behind `sub` from candidate h's exit state until the state it reaches
/// is one of the candidates of the subsequence it has just decoded (or that subsequence opens a restart segment, whose
/// hypothesis 0 is exact), kMhSteps at most.
template <int W, class JS>
__global__ __launch_bounds__(T) void huff_mh_flow(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const JobView J(js.get());
    ScanParams sp = J.sp;
    sp.use_sync_pack();
    load_tables(smem, J.tables_sync, sp);
    __syncthreads();
    const int sub = blockIdx.x * T + threadIdx.x, h = blockIdx.y;
    const int S = sp.num_subseq, H = sp.mh;
    if (sub >= S) return;
    const size_t at   = static_cast<size_t>(h) * S + sub;
    const Segment seg = ld_global(J.segments + J.seg_idx[sub]);
    const int seg_end = seg.subseq_offset + seg.subseq_count;
    LaneState st{};
    st.p = J.mh_p[at];
    {
        const int cz = J.mh_cz[at];
        st.c         = cz & 0xFF;
        st.z         = cz >> 8;
    }
    GlobalFetch<W> fetch{reinterpret_cast<JG_GLOBAL const uint32_t*>(J.destuffed), 0, 0};
    SpecSink none;
    uint32_t link = kMhNoLink, pool = kMhNoPool;
    const uint32_t pool_cap = mh_pool_entries(static_cast<uint32_t>(S));
    for (int k = 1; k <= kMhSteps; ++k) {
        const int t = sub + k;
        if (t >= S) break;       // the scan ends here
        if (t >= seg_end) {      // t opens the next segment: its hypothesis 0 starts in the true state
            link = static_cast<uint32_t>(k) | pool << 8;
            break;
        }
        const int rel = t - seg.subseq_offset;
        fetch.set_row(t, rel);
        BitWindow<GlobalFetch<W>> bw{};
        bw.seek(st.p, fetch);
        decode_subsequence(st, bw, fetch, (rel + 1) * (W * 32), smem, sp, none);
        const int cz = st.c | (st.z << 8);
        int g        = -1;
        for (int q = 0; q < H; ++q) {
            const size_t c = static_cast<size_t>(q) * S + t;
            if (J.mh_p[c] == st.p && J.mh_cz[c] == cz) g = q;
        }
        if (g >= 0) {
            link = static_cast<uint32_t>(k) | static_cast<uint32_t>(g) << 4 | pool << 8;
            break;
        }
        // no candidate of t is this state: keep it -- if this flow is the chain's, it is the true state of t
        if (k == 1) {
            const uint32_t first = __hip_atomic_fetch_add(reinterpret_cast<JG_GLOBAL uint32_t*>(J.mh_pool), static_cast<uint32_t>(kMhSteps - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (first + (kMhSteps - 1) <= pool_cap) pool = first;
        }
        if (pool != kMhNoPool && k < kMhSteps) st_global(J.mh_pool + 1 + pool + (k - 1), uint2_t{static_cast<uint32_t>(st.p), static_cast<uint32_t>(cz)});
    }
    J.mh_link[at] = link;
}

/// LDS of huff_mh_resolve for a segment of n subsequences with H candidates each.
__host__ __device__ constexpr size_t mh_resolve_lds(int H, int n)
{
    return static_cast<size_t>(H) * n * 8 + static_cast<size_t>(n) * 5 + static_cast<size_t>(n / 16 + 4) * 2 + 64;
}

// values of the successor tables (16 bits): a node h * n + r of the range at hand, or one of
constexpr uint32_t kMhEnd = 0xFFFFu, kMhBreak = 0xFFFEu; // the segment ends / the candidate met none within kMhSteps
constexpr uint32_t kMhExit = 0x8000u;                      // | h << 3 | row: the link leaves the BLOCK, into that node of the next one
static_assert(kMhMaxHyp * kMhMaxSegSubseq <= static_cast<int>(kMhExit) && kMhMaxHyp <= 8 && kMhSteps <= 8, "node ids, exit codes");

/// Successor of node (q, r) of the range [first, first + n) of a segment that ends at seg_end, from its link.
__device__ __forceinline__ uint32_t mh_successor(uint32_t l, int r, int n, int first, int seg_end)
{
    const int k = static_cast<int>(l & 15u), r2 = r + k;
    if (k == 0) return kMhBreak;
    if (first + r2 >= seg_end) return kMhEnd;
    const uint32_t g = (l >> 4) & 15u;
    return r2 < n ? g * static_cast<uint32_t>(n) + static_cast<uint32_t>(r2) : kMhExit | g << 3 | static_cast<uint32_t>(r2 - n);
}

/// Block-wise chain walk, step 1 (jg_defs.h): one workgroup per block; for each of the 64 nodes (h, row 0..7) the chain can
/// enter the block at, where it leaves it -- an exit code into the next block, the segment's end, or a break. Every node's
/// successor is squared ceil(log2 n) times (pointer jumping; exits, ends and breaks absorb).
template <class JS>
__global__ __launch_bounds__(256) void huff_mh_block_maps(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const JobView J(js.get());
    if (static_cast<int>(blockIdx.x) >= J.num_mh_blocks) return;
    const MhBlock blk = ld_global(J.mh_blocks + blockIdx.x);
    const int n = blk.count, first = blk.first, S = J.sp.num_subseq, H = J.sp.mh, nodes = H * n;
    uint16_t* s_a = reinterpret_cast<uint16_t*>(smem); // [nodes] (ping-pong)
    uint16_t* s_b = s_a + nodes;
    const int t   = threadIdx.x;
    for (int i = t; i < nodes; i += 256) {
        const int q = i / n, r = i - q * n;
        s_a[i]      = static_cast<uint16_t>(mh_successor(J.mh_link[static_cast<size_t>(q) * S + first + r], r, n, first, blk.seg_end));
    }
    __syncthreads();
    for (int span = 1; span < n; span <<= 1) {
        for (int i = t; i < nodes; i += 256) {
            const uint32_t a = s_a[i];
            s_b[i]           = static_cast<uint16_t>(a >= kMhExit ? a : s_a[a]);
        }
        __syncthreads();
        uint16_t* sw = s_a;
        s_a          = s_b;
        s_b          = sw;
    }
    if (t < 64) {
        const int h = t >> 3, r = t & 7;
        J.mh_blk_exit[static_cast<size_t>(blockIdx.x) * 64 + t] = static_cast<uint16_t>(h < H && r < n ? s_a[h * n + r] : kMhBreak);
    }
}

/// Step 2: one workgroup strings the maps together. A segment's first block is entered at node (0, 0) -- hypothesis 0 of a
/// segment's first subsequence is exact --, block b at what block b - 1 says for ITS entry node; behind a break every
/// block of the segment is marked kMhBlockBroken (plain speculation there).
template <class JS>
__global__ __launch_bounds__(256) void huff_mh_block_chain(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const JobView J(js.get());
    const int nb    = J.num_mh_blocks;
    uint16_t* s_map = reinterpret_cast<uint16_t*>(smem); // [nb][64]
    int* s_opens    = reinterpret_cast<int*>(s_map + static_cast<size_t>(nb) * 64);
    for (int i = threadIdx.x; i < nb * 64; i += 256) s_map[i] = J.mh_blk_exit[i];
    for (int i = threadIdx.x; i < nb; i += 256) s_opens[i] = ld_global(J.mh_blocks + i).opens;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t entry = 0;
        for (int b = 0; b < nb; ++b) {
            if (s_opens[b]) entry = 0;
            J.mh_blk_entry[b] = static_cast<uint16_t>(entry);
            if (entry == kMhBlockBroken) continue;
            const uint32_t x = s_map[b * 64 + entry];
            entry            = x == kMhBreak ? kMhBlockBroken : x == kMhEnd ? 0u : x & 63u; // (behind an end a segment opens)
        }
    }
}

/// The chain: one workgroup per restart segment (kBlocks: per block of the block-wise walk, from the entry node the two
/// kernels above found) loads the range's links into LDS and follows them from the entry node -- the first subsequence's
/// hypothesis 0 for a whole segment --; everybody then writes the table huff_sync_intra starts from
/// (in the rows of hypothesis 0 of mh_p / mh_cz): the candidate the chain
/// passes through where it does, the state a hopping flow left in the pool where the chain hopped over a subsequence
/// (both `mh_known`), hypothesis 0 as a placeholder where it hopped and the pool was full, and the plain speculation
/// from where the chain broke off (a candidate that met none within kMhSteps).
///
/// Followed link by link by one lane the chain of a 21 KB segment is 334 dependent LDS reads, 20 us for a kernel whose
/// whole point is a shorter critical path. So the links are squared four times first (every node gets its 16th
/// successor: pointer jumping, all lanes), one lane walks those -- n / 16 steps -- and leaves an ANCHOR every 16
/// links, and the lanes then walk the 16 links behind each anchor in parallel.
template <class JS, bool kBlocks>
__global__ __launch_bounds__(256) void huff_mh_resolve(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_broke, s_anchors; // first subsequence behind the point where the chain broke off (n: nowhere); anchors
    const JobView J(js.get());
    const ScanParams& sp = J.sp;
    int n, base, seg_end, row0 = 0; // the range, its segment's end, the row the chain enters it at
    uint32_t entry = 0;             // ... as a node
    if (kBlocks) {
        if (static_cast<int>(blockIdx.x) >= J.num_mh_blocks) return;
        const MhBlock blk = ld_global(J.mh_blocks + blockIdx.x);
        n       = blk.count;
        base    = blk.first;
        seg_end = blk.seg_end;
        entry   = J.mh_blk_entry[blockIdx.x];
    } else {
        if (static_cast<int>(blockIdx.x) >= sp.num_segments) return;
        const Segment seg = ld_global(J.segments + blockIdx.x);
        n       = seg.subseq_count;
        base    = seg.subseq_offset;
        seg_end = base + n;
    }
    const int S = sp.num_subseq, H = sp.mh;
    if (n <= 0) return; // (both walks refuse a segment without data; block-uniform, before anything is carved for n nodes)
    if (n > kMhMaxSegSubseq || entry == kMhBlockBroken) {
        // plain speculation: hypothesis 0's rows are the table (a segment too long for this kernel: only a device-scanned
        // image can get here, the host walk knows its segments and asks for blocks; a block behind a break)
        for (int r = threadIdx.x; r < n; r += 256) J.mh_known[base + r] = 1;
        return;
    }
    const int nodes  = H * n;                                               // node (h, r) = h * n + r
    uint32_t* s_link = reinterpret_cast<uint32_t*>(smem);                    // [nodes]
    uint32_t* s_src  = s_link + nodes;                                       // [n]: 1 + pool entry of a subsequence the chain hopped over, 0 none
    uint16_t* s_ja   = reinterpret_cast<uint16_t*>(s_src + n);               // [nodes] successor tables (ping-pong)
    uint16_t* s_jb   = s_ja + nodes;
    uint16_t* s_anc  = s_jb + nodes;                                         // [n / 16 + 2]: h << 12 | r of every 16th node of the chain
    uint8_t* s_hyp   = reinterpret_cast<uint8_t*>(s_anc + (n / 16 + 4));     // [n]: candidate the chain passes through, 0xFF none
    const int t = threadIdx.x;
    if (kBlocks) {
        row0  = static_cast<int>(entry & 7u);
        entry = (entry >> 3) * static_cast<uint32_t>(n) + static_cast<uint32_t>(row0);
    }
    for (int i = t; i < nodes; i += 256) {
        const int q = i / n, r = i - q * n;
        const uint32_t l = J.mh_link[static_cast<size_t>(q) * S + base + r];
        s_link[i]        = l;
        const uint32_t x = mh_successor(l, r, n, base, seg_end);
        s_ja[i]          = static_cast<uint16_t>(x >= kMhExit && x < kMhBreak ? kMhEnd : x); // (for the walk an exit is an end)
    }
    for (int r = t; r < n; r += 256) {
        s_hyp[r] = 0xFF;
        s_src[r] = 0;
    }
    if (t == 0) s_broke = n;
    __syncthreads();
    for (int round = 0; round < 4; ++round) { // successor -> 16th successor
        for (int i = t; i < nodes; i += 256) {
            const uint32_t a = s_ja[i];
            s_jb[i]          = static_cast<uint16_t>(a >= kMhBreak ? a : s_ja[a]);
        }
        __syncthreads();
        uint16_t* sw = s_ja;
        s_ja         = s_jb;
        s_jb         = sw;
    }
    if (t == 0) {
        int m = 0;
        uint32_t a = entry; // node (0, 0) of a segment: hypothesis 0 of its first subsequence is exact
        while (a < kMhBreak && m <= n / 16 + 1) { // (an anchor every 16 links of a chain of at most n: the bound is belt and braces)
            const uint32_t h = a / n;
            s_anc[m++]       = static_cast<uint16_t>(h << 12 | (a - h * n));
            a                = s_ja[a];
        }
        s_anchors = m;
    }
    __syncthreads();
    for (int i = t; i < s_anchors; i += 256) {
        int h = s_anc[i] >> 12, r = s_anc[i] & 0xFFF;
        for (int step = 0; step < 16; ++step) {
            s_hyp[r]         = static_cast<uint8_t>(h);
            const uint32_t l = s_link[h * n + r];
            const int k      = static_cast<int>(l & 15u);
            if (k == 0) {
                s_broke = r + 1;
                break;
            }
            const uint32_t pool = l >> 8;
            for (int q = 1; q < k && base + r + q < seg_end; ++q) {
                const uint32_t src = pool == kMhNoPool ? 0u : 1u + pool + static_cast<uint32_t>(q - 1);
                if (r + q < n) {
                    s_src[r + q] = src;
                } else {
                    // (block-wise walk) a subsequence of the NEXT block the chain hops over: this workgroup writes its
                    // table entry, the next block's leaves the rows in front of its entry row alone
                    const int sub = base + r + q;
                    if (src != 0) {
                        const uint2_t e = ld_global(J.mh_pool + src);
                        J.mh_p[sub]     = static_cast<int>(e.x);
                        J.mh_cz[sub]    = static_cast<int>(e.y);
                    }
                    J.mh_known[sub] = src != 0 ? 1 : 0;
                }
            }
            r += k;
            if (r >= n) break;
            h = static_cast<int>((l >> 4) & 15u);
        }
    }
    __syncthreads();
    // Behind a break the table is the reference's plain speculation (hypothesis 0, every lane flows): only entries the
    // chain HOPPED over may stay unknown -- at most kMhSteps - 1 in a row, right behind a lane that flows through them;
    // a longer run of entries nobody derives could leave the inter-sequence pass a stored state to stop at that nothing
    // downstream was derived from.
    for (int r = s_broke + t; r < n; r += 256) s_hyp[r] = 0;
    __syncthreads();
    for (int r = row0 + t; r < n; r += 256) { // (rows in front of the entry row belong to the previous block's workgroup)
        const int h = s_hyp[r];
        int p, cz;
        bool known = true;
        if (h != 0xFF) {
            const size_t at = static_cast<size_t>(h) * S + base + r;
            p               = J.mh_p[at];
            cz              = J.mh_cz[at];
        } else if (s_src[r] != 0) {
            const uint2_t e = ld_global(J.mh_pool + s_src[r]);
            p               = static_cast<int>(e.x);
            cz              = static_cast<int>(e.y);
        } else {
            p     = J.mh_p[base + r];
            cz    = J.mh_cz[base + r];
            known = false;
        }
        // The table lives in the rows of hypothesis 0, which nothing reads any more (entry base + r only by this lane,
        // above): huff_sync_intra starts from buffers no workgroup of ITS launch writes (st_p / st_cz, which round 3
        // used, are stored by the owners of the subsequences while the neighbours' overlap lanes read them: ADVICE r3).
        J.mh_p[base + r]     = p;
        J.mh_cz[base + r]    = cz;
        J.mh_known[base + r] = known ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: inter-sequence synchronisation
// ------------------------------------------------------------------------------------------------

/// Lanes of the tail kernel. A part cut from a scan with restart markers holds ~1000 subsequences (jg_defs.h) and
/// 60-80 flows, spread over the four waves of a 256-lane workgroup: idle waves of the ~200 us this latency-bound
/// kernel lives must not hold the wave slots other streams' kernels need (16-wave workgroups held 3/4 of them: -8 %
/// throughput with overlapping streams). A scan without restart markers is one part with thousands of flows: 1024
/// lanes, or its ordered groups of flows would run one after the other.
#ifndef JG_TAIL_LANES_SMALL
#define JG_TAIL_LANES_SMALL 256
#endif
#ifndef JG_TAIL_LARGE_FROM
#define JG_TAIL_LARGE_FROM 4096 // (1024-lane workgroups fit two to a CU: parts of ~1060 subsequences in them took 590 us per 64 images)
#endif
constexpr int kTailLanesSmall = JG_TAIL_LANES_SMALL, kTailLanesLarge = 1024, kTailLargeFrom = JG_TAIL_LARGE_FROM;

/// tails[b] (huff_seq_tails below: the sums of n and of the DC differences over the subsequences of sequence b that belong
/// to the segment still open at b's end) for the sequences whose LAST subsequence lies in [lo, hi): a part is a run of whole
/// segments, so everything such a sequence sums lies inside the part, whose states are final when its workgroup is done.
/// Folded into the tail kernel's workgroups (round 5): one launch and its gap fewer in every decode -- 8 of a lone 12 MP
/// decode's 245 us. Only for the 256-lane kernel: a part of a scan without restart markers holds hundreds of sequences,
/// which huff_seq_tails sums side by side.
template <int TL, bool kCoherent = false>
__device__ __forceinline__ void part_seq_tails(const JobView& J, int lo, int hi, uint32_t* s_red)
{
    static_assert(TL == T, "one sequence per round of the workgroup's lanes");
    const int SEQ = J.sp.seq_subseq, S = J.sp.num_subseq, t = threadIdx.x;
    for (int b = lo / SEQ; b * SEQ < S; ++b) {
        const int first = b * SEQ, last = min(first + SEQ, S) - 1;
        if (last < lo) continue; // (cannot happen: b starts at the sequence that holds lo)
        if (last >= hi) break;   // the sequence ends in a later part
        const int open_from = ld_global(J.segments + J.seg_idx[last]).subseq_offset;
        const int sub       = first + t;
        const bool take     = sub <= last && sub >= open_from;
        uint32_t n = 0u, d01 = 0u, d23 = 0u;
        if (take) {
            n   = static_cast<uint32_t>(ld_shared<kCoherent>(J.st_n + sub));
            d01 = ld_shared<kCoherent>(J.st_dc01 + sub);
            d23 = ld_shared<kCoherent>(J.st_dc23 + sub);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            n += __shfl_down(n, d);
            d01 = pk_add_u16(d01, __shfl_down(d01, d));
            d23 = pk_add_u16(d23, __shfl_down(d23, d));
        }
        __syncthreads(); // (the previous round's sums have been read)
        if (lane_id() == 0) {
            s_red[(t >> 6) * 3 + 0] = n;
            s_red[(t >> 6) * 3 + 1] = d01;
            s_red[(t >> 6) * 3 + 2] = d23;
        }
        __syncthreads();
        if (t == 0) {
            st_shared<kCoherent>(J.tails_n + b, static_cast<int>(s_red[0] + s_red[3] + s_red[6] + s_red[9]));
            st_shared<kCoherent>(J.tails_dc01 + b, pk_add_u16(pk_add_u16(s_red[1], s_red[4]), pk_add_u16(s_red[7], s_red[10])));
            st_shared<kCoherent>(J.tails_dc23 + b, pk_add_u16(pk_add_u16(s_red[2], s_red[5]), pk_add_u16(s_red[8], s_red[11])));
        }
    }
}

/// Continues, from global state, every flow that huff_sync_intra could not finish: one flow per
/// sequence boundary (carry the exit state of the last subsequence of sequence b-1 into sequence b,
/// b+1, ... until it meets the stored state or the segment ends) and one per pending mark. A flow
/// never leaves its segment, so the scan is cut at segment starts into parts that independent
/// workgroups own. Inside a part all flows advance in lock-step, one subsequence per iteration, and
/// groups of TL flows are processed in stream order, so a flow that started further upstream always
/// overwrites later and, unlike the reference (Appendix B-4), no pair of
/// flows is left unordered.
///
/// About half of the live flows synchronise in every iteration. The survivors are re-packed into the
/// lowest lanes after each iteration (their state is five words; the bit window is re-seeked from
/// `p`), so whole waves drop out and the cost follows the geometric decay instead of staying at one
/// full pass per iteration. Bitstream words come straight from the destuffed buffer, one refill
/// ahead; only the Huffman tables and the repacking buffer need LDS. With the overlap lanes of
/// huff_sync_intra a boundary flow normally confirms the stored state in its first iteration; this
/// kernel is what makes the result exact.
template <int W, int TL, bool kCoherent = false>
__device__ __forceinline__ void tail_part(const JobView& J, int part, uint8_t* smem, int* s_wave, uint32_t* s_red)
{
    constexpr bool kFusedTails = TL == T; // (the 1024-lane kernel of large parts leaves the sequence tails to huff_seq_tails)
    int* s_j       = reinterpret_cast<int*>(smem); // next subsequence of a surviving flow
    int* s_pz      = s_j + TL;                     // its bit position
    int* s_cz      = s_pz + TL;                    // its c | z << 8
    uint8_t* s_tab = reinterpret_cast<uint8_t*>(s_cz + TL);

#if !defined(JG_NO_TAIL_PRIO)
    // A part is a chain of dependent instructions of a few lanes: where its waves share a SIMD with the waves of throughput
    // kernels (other streams' launches; the writers of huff_tail_write) they go first -- what they ask of the issue slots is
    // next to nothing, and everything downstream waits for them.
    __builtin_amdgcn_s_setprio(3);
#endif
    ScanParams sp          = J.sp;
    sp.use_sync_pack();
    JG_GLOBAL const uint32_t* scan32 = reinterpret_cast<JG_GLOBAL const uint32_t*>(J.destuffed);
    const int lo           = J.tail_parts[part];
    const int hi           = J.tail_parts[part + 1];
    const int tid          = threadIdx.x;
    JG_TAIL_STAMP(0, static_cast<uint32_t>(wall_clock64()));

    // Ordered list of flow origins in [lo, hi): the marks huff_sync_intra left, and the sequence boundaries at which the
    // exit state the following sequence's workgroup assumed for its predecessor (bnd_p / bnd_cz) is not the one stored:
    // where it is, the sequence's first entry was derived from the stored state already and a flow could only confirm it.
    // (A decode that keeps every flow in its sequence's workgroup -- a lone decode: max_intra_iters is the lane count --
    // leaves no marks: only the last subsequence of every sequence is looked at, one round of the loop instead of one per
    // TL subsequences: 48 000 subsequences of a 12 MP scan without restart markers are ONE part, and building its list took
    // 60 of the kernel's 70 us.)
    // (ScanParams::tail_marks: set by build_jobs beside max_intra_iters -- up to round 4 this kernel inferred it from the cap)
    // CONTRACT with the sequence kernels (huff_sync_intra / huff_sync_intra_batch), which nothing else states:
    //   * tail_marks == 0: every flow was followed to its end inside its sequence's workgroup; only a flow that would cross
    //     into the next sequence is left, and bnd_p / bnd_cz say what that sequence's workgroup assumed in its place;
    //   * tail_marks != 0: `pending[i]` is set for every entry i a flow was cut short at;
    //   * either way a subsequence that OPENS a segment was decoded from the segment's start state (the flow pass's
    //     `covers_starts`, or an exact speculative pass): no flow below ever enters a segment from the one in front of it.
    const bool marks = sp.tail_marks != 0;
    const int SEQ    = sp.seq_subseq;
    const int step   = marks ? 1 : SEQ;
    int count = 0;
    for (int base = marks ? lo : (lo + SEQ) / SEQ * SEQ - 1; base < hi; base += TL * step) {
        const int sub = base + tid * step;
        bool f        = sub < hi && sub + 1 < sp.num_subseq;
        if (f) {
            f = J.pending[sub] != 0;
            if (!f && (sub + 1) % SEQ == 0 && J.seg_idx[sub] == J.seg_idx[sub + 1]) {
                const int b = (sub + 1) / SEQ;
                f           = J.bnd_p[b] != J.st_p[sub] || J.bnd_cz[b] != J.st_cz[sub];
            }
        }
        int rank;
        const int total = block_rank<TL>(f, s_wave, rank);
        if (f) J.flow_list[lo + count + rank] = sub;
        count += total;
    }
    if (count == 0) { // (uniform) most parts of a lone decode: nothing crosses a sequence boundary unconfirmed
        if constexpr (kFusedTails) part_seq_tails<TL, kCoherent>(J, lo, hi, s_red);
        return;
    }
    load_tables(s_tab, J.tables_sync, sp);
    __syncthreads(); // list and tables visible to the whole workgroup
    JG_TAIL_STAMP(1, static_cast<uint32_t>(wall_clock64()));
    [[maybe_unused]] int trips = 0;

    NoSink sink;
    for (int g = 0; g < count; g += TL) {
        // a flow between iterations: it has reached subsequence j - 1 with state (p, c, z)
        int j = 0, p = 0, cz = 0;
        bool live = false;
        // flow k of a group runs on lane (k % waves) * 64 + k / waves: a part's ~64 flows spread over the workgroup's waves,
        // a quarter of them on each of four SIMDs, instead of filling one wave (a trip takes as long as its slowest lane and
        // every step of a wave pays the LDS bank conflicts of all its lanes: 244 -> 231 us per 64 images, +1.2 % images/s)
        const int slot = (tid & 63) * (TL / 64) + (tid >> 6);
        if (g + slot < count) {
            const int from = J.flow_list[lo + g + slot];
            j              = from + 1;
            p              = J.st_p[from];
            cz             = J.st_cz[from];
            live           = true;
        }
        while (true) {
            bool flowing = false;
            if (live) {
                const Segment seg = ld_global(J.segments + J.seg_idx[j - 1]);
                const int lim     = seg.subseq_offset + seg.subseq_count;
                if (j < lim) {
                    GlobalFetch<W> fetch{scan32, 0, 0};
                    fetch.set_row(j, j - seg.subseq_offset);
                    LaneState st{};
                    st.p = p;
                    st.c = cz & 0xFF;
                    st.z = cz >> 8;
                    BitWindow<GlobalFetch<W>> bw{};
                    bw.seek(st.p, fetch);
                    decode_subsequence(st, bw, fetch, (j - seg.subseq_offset + 1) * (W * 32), s_tab, sp, sink);
                    p        = st.p;
                    cz       = st.c | (st.z << 8);
                    flowing  = !(p == J.st_p[j] && cz == J.st_cz[j]) && j + 1 < lim;
                    st_shared<kCoherent>(J.st_p + j, p);
                    st_shared<kCoherent>(J.st_n + j, st.n);
                    st_shared<kCoherent>(J.st_cz + j, cz);
                    st_shared<kCoherent>(J.st_dc01 + j, st.dc01);
                    st_shared<kCoherent>(J.st_dc23 + j, st.dc23);
                    ++j;
                }
            }
            // re-pack the survivors into the lowest lanes (also the barrier + workgroup-scope fence
            // that makes this iteration's state stores visible to the next one)
            int rank;
            const int total = block_rank<TL>(flowing, s_wave, rank);
            if (trips < 30) {
                JG_TAIL_STAMP(2 + trips, static_cast<uint32_t>(wall_clock64()));
                JG_TAIL_STAMP(32 + trips, static_cast<uint32_t>(__syncthreads_count(live))); // flows of the trip
            }
            ++trips;
            if (total == 0) break;
            if (flowing) {
                s_j[rank]  = j;
                s_pz[rank] = p;
                s_cz[rank] = cz;
            }
            __syncthreads();
            live = slot < total;
            if (live) {
                j  = s_j[slot];
                p  = s_pz[slot];
                cz = s_cz[slot];
            }
        }
        __syncthreads();
    }
    JG_TAIL_STAMP(63, static_cast<uint32_t>(count) | static_cast<uint32_t>(trips) << 16);
    if constexpr (kFusedTails) {
        __syncthreads(); // every flow's stores are done and visible to the workgroup
        part_seq_tails<TL, kCoherent>(J, lo, hi, s_red);
    }
}

template <int W, int TL, class JS>
__global__ __launch_bounds__(TL) void huff_sync_tail(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_wave[TL / 64];
    __shared__ uint32_t s_red[12];
    const JobView J(js.get());
    if (static_cast<int>(blockIdx.x) >= J.num_tail_parts) return;
    tail_part<W, TL>(J, static_cast<int>(blockIdx.x), smem, s_wave, s_red);
}

// ------------------------------------------------------------------------------------------------
// Huffman: per-sequence tails
// ------------------------------------------------------------------------------------------------

template <bool kPacked>
__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* s_red)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = __shfl_down(v, d);
        v                = kPacked ? pk_add(v, o) : v + o;
    }
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (kPacked) return pk_add(pk_add(s_red[0], s_red[1]), pk_add(s_red[2], s_red[3]));
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

/// tails[b] = sum of (n, dc) over the subsequences of sequence b that belong to the segment still
/// open at the end of b. The write pass of sequence b' > b in the same segment adds tails[a..b'-1]
/// (a = sequence holding the segment's first subsequence) to get its offset inside the segment.
template <class JS>
__global__ __launch_bounds__(T) void huff_seq_tails(JS js)
{
    __shared__ uint32_t s_red[4];
    const JobView J(js.get());
    if (static_cast<int>(blockIdx.x) >= J.num_seq) return;
    const int t         = threadIdx.x;
    const int SEQ       = J.sp.seq_subseq;
    const int first_sub = blockIdx.x * SEQ;
    const int nsub      = min(SEQ, J.sp.num_subseq - first_sub);
    const int last_seg  = J.seg_idx[first_sub + nsub - 1];
    const int open_from = J.segments[last_seg].subseq_offset; // global index of that segment's start
    const int sub       = first_sub + t;
    const bool take     = t < nsub && sub >= open_from;
    const uint32_t n    = block_sum_256<false>(take ? J.st_n[sub] : 0, s_red);
    const uint32_t d01  = block_sum_256<true>(take ? J.st_dc01[sub] : 0u, s_red);
    const uint32_t d23  = block_sum_256<true>(take ? J.st_dc23[sub] : 0u, s_red);
    if (t == 0) {
        J.tails_n[blockIdx.x]    = static_cast<int>(n);
        J.tails_dc01[blockIdx.x] = d01;
        J.tails_dc23[blockIdx.x] = d23;
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman: write pass
// ------------------------------------------------------------------------------------------------

constexpr int kRingWords    = 16;                         // 32-bit words of a lane's write-combining ring: two entries each
constexpr int kRingStride   = (kRingWords + 1) * 4;       // bytes from one lane's ring to the next: 17 words, an odd number of banks
constexpr int kStageEntries = 2 * kRingWords;             // entries the ring holds (a power of two)
constexpr int kFlushEntries = kSymSectorEntries;          // entries per flush: 16 = one 32-byte sector
#ifndef JG_WRITE_FLUSH_PERIOD
#define JG_WRITE_FLUSH_PERIOD 12
#endif
constexpr int kWriteFlushPeriod = JG_WRITE_FLUSH_PERIOD;  // iterations between two flush points
// what a flush point leaves behind (less than a sector) + what arrives until the next one (one entry per iteration and
// one per DC slot; an ESCAPE entry, which no photograph has, flushes for itself when a sector is waiting: one more)
// must fit the ring
static_assert(kFlushEntries - 1 + kWriteFlushPeriod + (kWriteFlushPeriod + kWriteDcPeriod - 1) / kWriteDcPeriod + 1 <= kStageEntries, "the write-combining ring would overflow");
#ifndef JG_UNIT_TURN
#define JG_UNIT_TURN 1
#endif
constexpr int kUnitTurn = JG_UNIT_TURN;                   // flush points between two stores of data-unit records (a power of two)
// A DC slot adds at most one waiting record, a turn leaves at most three, StreamSink holds eight.
static_assert(3 + kUnitTurn * ((kWriteFlushPeriod + kWriteDcPeriod - 1) / kWriteDcPeriod) <= 8, "waiting unit records would overflow");

/// Sink of the write pass: a compact symbol stream instead of a dense coefficient buffer (jg_defs.h). Every lane
/// appends 16-bit entries to its own region, contiguous per data unit, and records {first entry, count} per
/// data unit when the unit completes. A lane owns whole data units (jg_huff_core.h, decode_units), so a unit's
/// entries never span two regions. Compared with scattered 2-byte stores into a pre-zeroed buffer (reference
/// decode_huffman.cu:360-371 + decoder.cpp:256-263) this needs no zero-fill and writes each byte once.
///
/// A lane's appends must not go to memory one by one: with many images in flight the ~200 k open lines do not fit
/// in L2 and every append becomes its own 32-byte sector write. Entries are therefore collected in a ring per lane
/// in LDS (32 entries = 64 bytes, rings 17 words apart: the lanes of a wave hit different banks) and every 12
/// iterations ALL lanes that have 16 or more waiting flush one whole 32-byte sector: an iteration adds one entry from
/// its AC section and every fourth one more from its DC slot, so at most 15 stay behind and at most 15 arrive in
/// between (an escape entry, the second of its iteration, flushes for itself if a sector is waiting). The period is as
/// long as the ring allows: in a wave of 64 some lane has a sector waiting at every flush point, and what the wave
/// pays is how often it enters that code (4 / 6 / 7 / 8 / 10 / 12 iterations: 800 / 791 / 790 / 780 / 785 / 770 µs).
///
/// The kernel is bound by vector-instruction issue, so the per-symbol part is counted in instructions: an AC symbol
/// is one ring store and an add-with-carry (zero coefficients, and everything a lane decodes before its first DC
/// symbol -- the tail of its predecessor's unit --, are stored where the next kept entry will overwrite them: no
/// select, no test; the first DC symbol rewinds the lane's count). The region cannot overflow on a valid stream
/// (jg_defs.h, sym_region_entries); on a corrupt one the flush drops what lies beyond it.
struct StreamSink {
    static constexpr int kFlushPeriod = kWriteFlushPeriod;
    JG_GLOBAL uint16_t* sym;
    JG_GLOBAL uint2_t* du_tab;
    uint32_t ring;      // LDS byte address of the lane's ring: entry n at ring + (n % 32) * 2
    uint32_t base;      // physical index of the region's first entry (jg_defs.h, sym_region_base)
    uint32_t flushed;   // entries of this lane already in memory (a multiple of 16), region-relative like the next three
    uint32_t emitted;   // entries produced so far
    uint32_t cur_end;   // entries a region holds
    uint32_t du_off;    // first entry of the unit being decoded, minus kUnitHasEscape once it has taken an escape entry
    int du;       // next data unit this lane starts
    int quota;    // first data unit past the segment
    // Data-unit records {first entry, count} wait here until FOUR of them fill a 32-byte sector of the table: a
    // lane's units are consecutive in the table and contiguous in its region, so the table index and the offset of
    // the first waiting one, plus packed entry counts (up to eight, one byte each), describe them. Stored one by
    // one as they completed, every 8-byte record became a memory write of its own -- the line it belongs to takes a
    // lane ~16 units to fill and does not live that long in L2: 11.5 MB of HBM writes per 12 MP image for a 2.3 MB
    // table (rocprofv3 WRITE_SIZE).
    uint32_t rec_off;   // region-relative offset of the first unit whose record has not been stored
    int rec_du;         // its index in the data-unit table
    uint32_t pend_lo, pend_hi; // counts (| kUnitHasEscape) of the waiting units: the TOP pend_n bytes of hi:lo, oldest lowest
    int pend_n;
    uint32_t started; // 0 until the lane's first DC symbol: what it decodes before finishes the predecessor's data unit
    __device__ __forceinline__ bool full() const { return du >= quota; }
    /// 16-bit store into the lane's ring, slot n % 32.
    __device__ __forceinline__ void put(uint32_t n, uint32_t entry)
    {
        typedef __attribute__((address_space(3))) uint16_t LdsHalf;
        uint32_t a; // ring + (n % 32) * 2 as AND + shift-add (the compiler turns it into shift, AND, add)
        asm("v_and_b32 %0, %1, %2\n\tv_lshl_add_u32 %0, %0, 1, %3" : "=&v"(a) : "n"(kStageEntries - 1), "v"(n), "v"(ring));
        *reinterpret_cast<LdsHalf*>(static_cast<uintptr_t>(a)) = static_cast<uint16_t>(entry);
    }
    /// DC slot, a lane at the start of a data unit: the unit it finished since the previous slot (if it has started
    /// one) is complete -- its entry count joins the waiting records at the top. At most one per slot, three between
    /// flush points; every kUnitTurn-th flush point leaves at most three waiting: eight places are enough.
    __device__ __forceinline__ void unit_boundary()
    {
        if (started) {
            const uint32_t cnt = emitted - du_off; // entries | kUnitHasEscape
            pend_lo            = __builtin_amdgcn_alignbit(pend_hi, pend_lo, 8);
            pend_hi            = __builtin_amdgcn_alignbit(cnt, pend_hi, 8);
            ++pend_n;
        }
    }
    /// The DC symbol that opens a unit: its absolute DC value is the unit's first entry.
    __device__ __forceinline__ void dc(int value)
    {
        emitted = started ? emitted : 0u; // what was stored before belongs to the predecessor's lane
        started = 1u;
        du_off  = emitted;
        ++du;
        put(emitted, static_cast<uint32_t>(value));
        ++emitted;
    }
    /// An AC symbol of category `category` whose coefficient `value` sits at zig-zag index `zpos`; category 0 (a run
    /// of zeros, an end of block, the null entry of a lane that does not step) keeps nothing.
    __device__ __forceinline__ void ac(int category, int zpos, int value)
    {
        put(emitted, sym_entry_ac(zpos, value));
        emitted += category != 0 ? 1u : 0u;
    }
    /// A coefficient of category 10 or more (no photograph has one) takes a second entry behind the one ac() stored.
    __device__ __forceinline__ void escape(int value)
    {
        put(emitted, sym_entry_escape(value));
        ++emitted;
        du_off -= ((emitted - du_off) & kUnitHasEscape) ? 0u : kUnitHasEscape; // once per unit (a unit has at most 127 entries)
        // the flush points are spaced for one entry per iteration: this second one makes room for itself
        if (started && emitted - flushed >= static_cast<uint32_t>(kFlushEntries)) flush_sector();
    }
    /// Store the first `n` waiting records (1..4).
    __device__ __forceinline__ void store_units(int n)
    {
        // the waiting counts, oldest in byte 0
        const uint32_t counts = static_cast<uint32_t>(((static_cast<uint64_t>(pend_hi) << 32) | pend_lo) >> (8 * (8 - pend_n)));
        uint2_t rec[4];
        uint32_t off = rec_off;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t cnt = (counts >> (8 * k)) & 0xFFu; // entries | kUnitHasEscape
            rec[k]             = uint2_t{sym_at(base, off), cnt};
            off += k < n ? cnt & 0x7Fu : 0u;
        }
        JG_GLOBAL uint2_t* dst = du_tab + rec_du;
#if defined(JG_EXP_NO_UNIT_STORES) // traffic experiments only (tools/probe/pmc_lib.sh): results are wrong
        if (rec[0].x == 0xFFFFFFFEu)
#endif
        if (n == 4 && (rec_du & 3) == 0) { // a whole, aligned sector of the table
            st_global(reinterpret_cast<JG_GLOBAL uint4*>(dst), make_uint4(rec[0].x, rec[0].y, rec[1].x, rec[1].y));
            st_global(reinterpret_cast<JG_GLOBAL uint4*>(dst) + 1, make_uint4(rec[2].x, rec[2].y, rec[3].x, rec[3].y));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n) st_global(dst + k, rec[k]);
        }
        rec_off = off;
        rec_du += n;
        pend_n -= n; // the others are still the top pend_n bytes
    }
    __device__ __forceinline__ void flush_units()
    {
        // up to the next sector boundary of the table; from then on the lane stores whole sectors (a second round
        // only right after the lane's first, shorter store: at most three records stay behind)
        while (pend_n >= 4) store_units(4 - (rec_du & 3));
    }
    /// One sector (16 entries = 8 ring words) from the ring to memory; `flushed` is a multiple of 16.
    __device__ __forceinline__ void flush_sector()
    {
        typedef __attribute__((address_space(3))) uint32_t LdsWord;
        uint32_t e[kFlushEntries / 2];
        const uint32_t r = ring + ((flushed & (kStageEntries - 1u)) << 1); // lower or upper half of the ring
#pragma unroll
        for (int k = 0; k < kFlushEntries / 2; ++k) e[k] = *reinterpret_cast<const LdsWord*>(static_cast<uintptr_t>(r + k * 4));
        static_assert(kFlushEntries == 16, "a flush is one sector of the interleaved stream");
        JG_GLOBAL uint4* dst = reinterpret_cast<JG_GLOBAL uint4*>(sym + sym_at(base, flushed));
#if defined(JG_EXP_NO_SECTOR_STORES)
        if (e[0] == 0xFFFFFFFEu && e[1] == 0x12345678u)
#endif
        if (flushed < cur_end) { // a region cannot overflow on a valid stream; a corrupt one loses what lies beyond it
            st_global(dst, make_uint4(e[0], e[1], e[2], e[3]));
            st_global(dst + 1, make_uint4(e[4], e[5], e[6], e[7]));
        }
        flushed += kFlushEntries;
    }
    /// Every kFlushPeriod-th iteration, the same one for every lane of the wave; `no` counts them. The unit records
    /// take every kUnitTurn-th.
    __device__ __forceinline__ void flush_point(int no)
    {
        if (started && emitted - flushed >= static_cast<uint32_t>(kFlushEntries)) flush_sector();
        if ((no & (kUnitTurn - 1)) == 0) flush_units();
    }
    /// After the loop: everything that is left, rounded up to whole sectors (the entries behind the
    /// last valid one are never read: the data-unit table bounds every gather).
    __device__ __forceinline__ void finish()
    {
        if (!started) return; // the lane's subsequence lies inside one data unit: nothing of it is kept here
        while (flushed < emitted) flush_sector();
        while (pend_n > 0) store_units(pend_n < 4 ? pend_n : 4);
    }
};

/// Exclusive prefix over the 256 lanes of `v` (plain, not segmented), result left in s_scan[0..T].
template <bool kPacked>
__device__ __forceinline__ void block_excl_scan_256(uint32_t v, uint32_t* s_scan, uint32_t* s_wave)
{
    const int t = threadIdx.x;
    uint32_t in = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(in, d);
        if (lane_id() >= d) in = kPacked ? pk_add(in, o) : in + o;
    }
    __syncthreads(); // previous use of s_scan / s_wave is over
    if (lane_id() == 63) s_wave[t >> 6] = in;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = k < (t >> 6) ? s_wave[k] : 0u;
        off              = kPacked ? pk_add(off, w) : off + w;
    }
    const uint32_t incl = kPacked ? pk_add(in, off) : in + off;
    // exclusive = inclusive - v, per 16-bit half when packed
    s_scan[t] = kPacked ? pk_add(incl, pk_add(~v, 0x00010001u)) : incl - v;
    if (t == T - 1) s_scan[T] = incl;
    __syncthreads();
}

__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_add(a, pk_add(~b, 0x00010001u)); }

/// Carve of the write kernel's dynamic LDS: scan scratch | write-combining rings | table pack.
struct WriteLds {
    static constexpr uint32_t kScan = 0;                                   // T + 1 + 4 + 3 words
    static constexpr uint32_t kRing = ((T + 8) * 4 + 15) / 16 * 16;
    static constexpr uint32_t kTabs = (kRing + kRingStride * kMaxSeq + 15) / 16 * 16; // lanes SEQ..T-1 emit nothing (SEQ <= 255)
    static_assert(kTabs % 16 == 0, "the table pack is read with 16-byte loads");
    static_assert(kStaticLdsSlack + kTabs + kMaxTablePack <= 65536, "absolute table addresses are packed into 16 bits (load_tables)");
};
static_assert(kStaticLdsSlack + 3 * kTailLanesLarge * 4 + kMaxTablePackSync <= 65536, "huff_sync_tail: the same bound");

/// Re-decode every subsequence from its predecessor's synchronised exit state and emit the symbol
/// stream (StreamSink). The coefficient-slot position of subsequence i inside its segment = sum of n
/// over the segment's earlier subsequences (in-sequence part by an LDS scan, earlier sequences via
/// tails); it gives the index of every data unit the lane starts, and the same look-back gives the
/// DC predictors. Lanes SEQ..T-1 have no subsequence here (the sequence is SEQ long); they only help
/// with the scans. The bitstream is read straight from the destuffed buffer, one refill ahead (as in
/// huff_sync_tail); LDS holds the ring and the tables: five workgroups per CU.
template <int W, bool kCoherent = false>
__device__ __forceinline__ void write_sequence(const JobView& J, int seq, uint8_t* smem)
{
    uint32_t* s_scan  = reinterpret_cast<uint32_t*>(smem + WriteLds::kScan); // T + 1
    uint32_t* s_wave  = s_scan + T + 1;                                       // 4
    uint32_t* s_carry = s_wave + 4;                                           // 3
    uint32_t* s_ring  = reinterpret_cast<uint32_t*>(smem + WriteLds::kRing);
    uint8_t* s_tab    = smem + WriteLds::kTabs;

    ScanParams sp = J.sp;
    const int t         = threadIdx.x;
    const int SEQ       = sp.seq_subseq;
    const int first_sub = seq * SEQ;
    const int nsub      = min(SEQ, sp.num_subseq - first_sub);

    // What the lane needs of its own subsequence and of the one in front, asked for before anything else and without
    // a branch (a lane without a subsequence reads the last one's): these loads travel while the table pack is copied.
    // Where each sat in front of its first use -- behind the scans' barriers, one after the other -- the workgroup
    // waited out seven memory latencies in a row before its first symbol.
    const int sub_c        = first_sub + min(t, nsub - 1);
    const int lane_seg     = J.seg_idx[sub_c];
    const int first_seg    = J.seg_idx[first_sub];
    const uint32_t lane_n  = static_cast<uint32_t>(ld_shared<kCoherent>(J.st_n + sub_c));
    const uint32_t lane_01 = ld_shared<kCoherent>(J.st_dc01 + sub_c), lane_23 = ld_shared<kCoherent>(J.st_dc23 + sub_c);
    const int prev_p       = ld_shared<kCoherent>(J.st_p + max(sub_c - 1, 0));
    const int prev_cz      = ld_shared<kCoherent>(J.st_cz + max(sub_c - 1, 0));

    load_tables(s_tab, J.tables, sp);
    const Segment lane_segment = ld_global(J.segments + lane_seg);

    // carry-in of the segment that is open at the sequence's first subsequence
    {
        const Segment seg0 = ld_global(J.segments + first_seg);
        const int a        = seg0.subseq_offset / SEQ; // sequence holding the segment's start
        uint32_t cn = 0, c01 = 0, c23 = 0;
        if (seg0.subseq_offset < first_sub) {
            for (int b = a + t; b < seq; b += T) {
                cn += static_cast<uint32_t>(ld_shared<kCoherent>(J.tails_n + b));
                c01 = pk_add(c01, ld_shared<kCoherent>(J.tails_dc01 + b));
                c23 = pk_add(c23, ld_shared<kCoherent>(J.tails_dc23 + b));
            }
        }
        cn  = block_sum_256<false>(cn, s_wave);
        c01 = block_sum_256<true>(c01, s_wave);
        c23 = block_sum_256<true>(c23, s_wave);
        if (t == 0) {
            s_carry[0] = cn;
            s_carry[1] = c01;
            s_carry[2] = c23;
        }
    }
    __syncthreads();

    const bool active = t < nsub;
    const int sub     = first_sub + t;
    int seg_i = 0, rel = 0, ts = 0; // ts = local index of the segment's first subsequence (clamped to 0)
    Segment seg{0, 0};
    bool carried = false; // segment started before this sequence
    if (active) {
        seg_i   = lane_seg;
        seg     = lane_segment;
        rel     = sub - seg.subseq_offset;
        carried = seg.subseq_offset < first_sub;
        ts      = carried ? 0 : seg.subseq_offset - first_sub;
    }

    StreamSink sink;
    sink.sym     = J.sym;
    sink.du_tab  = J.du_tab;
    sink.ring    = lds_address(s_ring) + static_cast<uint32_t>(t) * kRingStride;
    int nprefix  = 0, nnext = 0; // coefficient slots of the segment in front of this lane's subsequence / of the next one's
    uint32_t pred01 = 0, pred23 = 0; // DC predictors at the lane's first symbol: sums over the segment so far
    {
        block_excl_scan_256<false>(active ? lane_n : 0u, s_scan, s_wave);
        nprefix = static_cast<int>(s_scan[t] - s_scan[ts] + (carried ? s_carry[0] : 0u));
        nnext   = static_cast<int>(s_scan[t + 1] - s_scan[ts] + (carried ? s_carry[0] : 0u));
        block_excl_scan_256<true>(active ? lane_01 : 0u, s_scan, s_wave);
        const uint32_t p01 = pk_add(pk_sub(s_scan[t], s_scan[ts]), carried ? s_carry[1] : 0u);
        block_excl_scan_256<true>(active ? lane_23 : 0u, s_scan, s_wave);
        const uint32_t p23 = pk_add(pk_sub(s_scan[t], s_scan[ts]), carried ? s_carry[2] : 0u);
        pred01 = p01;
        pred23 = p23;
    }
    if (!active) return;

    const int seg_mcus0 = seg_i * sp.mcus_per_segment;
    const int seg_mcus1 = min(seg_mcus0 + sp.mcus_per_segment, sp.total_mcus); // Appendix B-5 clamp
    sink.du             = seg_mcus0 * sp.du_per_mcu + ((nprefix + 63) >> 6);
    // The lane stops in front of the first unit the NEXT lane starts (the same formula one subsequence on: a lane owns
    // the units whose DC symbol its subsequence commits, reference decode_huffman.cu:350-355), or at the segment's
    // quota of data units if it is the segment's last (Appendix B-5).
    sink.quota          = seg_mcus1 * sp.du_per_mcu;
    if (rel + 1 < seg.subseq_count) sink.quota = min(sink.quota, seg_mcus0 * sp.du_per_mcu + ((nnext + 63) >> 6));
    sink.du = min(sink.du, sink.quota); // both follow from consistent counts; should they ever not, nothing leaves the table
    sink.base           = sym_region_base(static_cast<uint32_t>(sub), J.sym_region);
    sink.flushed        = 0;
    sink.emitted        = 0;
    sink.cur_end        = J.sym_region;
    sink.du_off         = 0;
    sink.rec_off        = 0;
    sink.rec_du         = sink.du; // the first unit this lane starts
    sink.pend_lo        = 0;
    sink.pend_hi        = 0;
    sink.pend_n         = 0;

    LaneState st{};
    st.dc01 = pred01;
    st.dc23 = pred23;
    if (rel > 0) {
        st.p         = prev_p;
        const int cz = prev_cz;
        st.c         = cz & 0xFF;
        st.z         = cz >> 8;
    }
    sink.started = 0u; // until the lane's first DC symbol
    RowWindow<W> words{GlobalFetch<W, true>{uniform_ptr(reinterpret_cast<JG_GLOBAL const uint32_t*>(J.destuffed)), 0, 0}};
    words.g.set_row(sub, rel);
    // Iterations a valid stream can need for the bits of the subsequence and of the unit the lane runs on into (64 symbols of
    // at most 32 bits): the densest stream is two-bit data units (a one-bit DC code of category 0 and a one-bit end of
    // block), and such a unit takes one DC slot and the AC step behind it -- kWriteDcPeriod iterations per two bits; a
    // symbol that waits for the rare slot has eleven bits or more and waits at most kWriteRarePeriod iterations.
    constexpr int kItersPerBitX2 = kWriteDcPeriod > 2 ? kWriteDcPeriod : 2; // iterations per TWO bits, worst case
    constexpr int kMaxIters      = kItersPerBitX2 * (W * 32 + 64 * 32) / 2 + 2 * kWriteRarePeriod;
    static_assert(kWriteRarePeriod <= 11, "a symbol that waits for the rare slot must not wait longer than its bits allow for");
#if defined(JG_PROBE)
    int iters[4] = {0, 0, 0, 0};
    decode_units(st, words, s_tab, sp, sink, kMaxIters, iters);
    atomicAdd(&g_probe_write[0], static_cast<unsigned long long>(iters[0]));
    atomicMax(&g_probe_write[2], static_cast<unsigned long long>(iters[0]));
    atomicAdd(&g_probe_write[3], 1ull);
    if (blockIdx.y == 0 && sub < (1 << 16)) {
        g_probe_lane_iters[sub]             = static_cast<uint16_t>(iters[0]);
        g_probe_lane_iters[(1 << 16) + sub] = static_cast<uint16_t>(iters[1]);
        if (sub < (1 << 15)) g_probe_lane_rare[sub] = static_cast<uint32_t>(iters[2]) | static_cast<uint32_t>(iters[3]) << 16;
    }
#else
    decode_units(st, words, s_tab, sp, sink, kMaxIters);
#endif
    sink.finish();
}

template <int W, class JS>
__global__ __launch_bounds__(T) void huff_write(JS js)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const JobView J(js.get());
    if (static_cast<int>(blockIdx.x) >= J.num_seq) return;
    write_sequence<W>(J, static_cast<int>(blockIdx.x), smem);
}

/// The tail kernel's parts AND the write pass's sequences of a batch as ONE launch (round 5). As two launches the chip
/// waits for the slowest part's chain of dependent flow iterations -- 2-4 whole subsequences by a handful of lanes, 200-500
/// us per 64 images during which 7 of 8 wave slots are empty -- before the first symbol is written. Here a workgroup takes
/// a TICKET when it starts (one atomic counter per launch) and the ticket is its role: first every part of every job, then
/// as many writers as there are sequences. A part that is done counts itself off every sequence it holds subsequences of
/// (a part is a run of whole segments: the states a sequence reads, its predecessor's exit state and the sequence tails its
/// look-back adds up -- part_seq_tails -- all lie in the parts that hold its subsequences) and pushes the sequences that
/// reach zero onto its job's READY QUEUE; a writer claims the next entry of a queue that has one, whichever job's: the
/// sequences of the parts that are done are written while the slow parts still run, and no writer sits on a wave slot
/// waiting for ITS sequence while others are ready (with sequences bound to writers by ticket the launch took as long as
/// the two kernels it replaces).
/// No deadlock by construction: a writer holds a ticket behind every part's, so every part HAS STARTED (is resident or
/// done) when a writer first waits, parts wait for nothing, and every sequence is pushed exactly once. Tickets rather
/// than blockIdx: the order in which workgroups start is not specified. The wait is bounded all the same (a queue that
/// never fills would hang the chip for every process on it): a writer that gives up leaves its sequence unwritten and
/// says so in g_fuse_timeouts.
/// What parts hand to writers goes through st_shared / ld_shared (every XCD has its own L2), the control words through
/// agent-scope atomics; a part waits for its stores to be taken (s_waitcnt vmcnt(0)) before it counts itself off.
__device__ unsigned int g_fuse_timeouts;
constexpr uint32_t kFuseMaxPolls = 1u << 23; // 10-25 s of polling: far beyond any part of a valid or a corrupt stream, even with the chip shared

/// `jobs[i]` as memory nothing writes while the kernel runs (the constant address space): behind the ticket's atomic the
/// compiler no longer takes the job array for unchanged and would load every field of the job per lane, into vector
/// registers -- the kernels keep a job's forty pointers and its parameters in scalar registers.
__device__ __forceinline__ const ScanJob& constant_job(const ScanJob* jobs, uint32_t i)
{
    typedef const __attribute__((address_space(4))) ScanJob* ConstJob;
    return *(const ScanJob*)(ConstJob)(jobs + i);
}

#define JG_FUSE_LOAD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define JG_FUSE_ADD(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define JG_FUSE_STORE(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

template <int W>
__global__ __launch_bounds__(T) void huff_tail_write(const ScanJob* jobs, int num_jobs, int max_parts, int max_seq)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_wave[T / 64];
    __shared__ uint32_t s_red[12];
    __shared__ uint32_t s_role, s_claim[2];
    JG_GLOBAL uint32_t* ctl0 = as_global(jobs[0].fuse_ctl);
    if (threadIdx.x == 0) s_role = JG_FUSE_ADD(ctl0, 1u);
    __syncthreads();
    // (uniform, and known to be: the job's pointers then live in scalar registers as in every other kernel)
    const uint32_t role = __builtin_amdgcn_readfirstlane(s_role), num_tail = static_cast<uint32_t>(num_jobs) * static_cast<uint32_t>(max_parts);
    if (role < num_tail) {
        const uint32_t j   = role / static_cast<uint32_t>(max_parts);
        const ScanJob& job = constant_job(jobs, j);
        const int part     = static_cast<int>(role % static_cast<uint32_t>(max_parts));
        const JobView J(job);
        if (part >= J.num_tail_parts) return;
        tail_part<W, T, true>(J, part, smem, s_wave, s_red);
        // this lane's stores (states, sequence tails: st_shared) have been taken by the memory every XCD reads; then the
        // part counts itself off its sequences and queues the ones it was the last holder of
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        JG_GLOBAL uint32_t* own = as_global(job.fuse_ctl);
        const int SEQ = J.sp.seq_subseq, lo = J.tail_parts[part], hi = J.tail_parts[part + 1];
        for (int q = lo / SEQ + static_cast<int>(threadIdx.x); q <= (hi - 1) / SEQ && q < J.num_seq; q += T) {
            if (JG_FUSE_ADD(own + kFuseOwn + q, 0xFFFFFFFFu) == 1u) {
                const uint32_t i = JG_FUSE_ADD(ctl0 + kFuseJobs + 2 * j, 1u);
                if (i < static_cast<uint32_t>(J.num_seq)) JG_FUSE_STORE(own + kFuseOwn + J.num_seq + i, static_cast<uint32_t>(q));
            }
        }
        return;
    }
    // a writer: one per sequence of every job (the static share only says whether there is one for this workgroup)
    const uint32_t r = role - num_tail;
    if (r >= static_cast<uint32_t>(num_jobs) * static_cast<uint32_t>(max_seq)) return;
    const uint32_t home = r / static_cast<uint32_t>(max_seq);
    if (static_cast<int>(r % static_cast<uint32_t>(max_seq)) >= constant_job(jobs, home).num_seq) return;
    if (threadIdx.x < 64) { // the first wave looks for a queue with an entry nobody has claimed, 64 jobs at a time from its own on
        uint32_t polls = 0, got_job = kFuseEmpty, got_i = 0;
        while (got_job == kFuseEmpty) {
            for (int base = 0; base < num_jobs && got_job == kFuseEmpty; base += 64) {
                const int k      = base + static_cast<int>(threadIdx.x);
                const uint32_t j = (home + static_cast<uint32_t>(k)) % static_cast<uint32_t>(num_jobs);
                bool has         = false;
                if (k < num_jobs) {
                    const uint32_t pushed = JG_FUSE_LOAD(ctl0 + kFuseJobs + 2 * j), claimed = JG_FUSE_LOAD(ctl0 + kFuseJobs + 2 * j + 1);
                    has                   = claimed < pushed;
                }
                const unsigned long long any = __builtin_amdgcn_ballot_w64(has);
                if (any) {
                    const int first = __builtin_ctzll(any);
                    const uint32_t jj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(j), first));
                    uint32_t i = 0;
                    if (threadIdx.x == 0) i = JG_FUSE_ADD(ctl0 + kFuseJobs + 2 * jj + 1, 1u);
                    i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(i)));
                    // (claimed beyond what is pushed yet: the entry will come; beyond the job's sequences: another writer was faster)
                    if (i < static_cast<uint32_t>(constant_job(jobs, jj).num_seq)) {
                        got_job = jj;
                        got_i   = i;
                    }
                }
            }
            if (got_job == kFuseEmpty) {
                if (++polls > kFuseMaxPolls) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
        uint32_t q = kFuseEmpty;
        if (got_job != kFuseEmpty) {
            const ScanJob& job = constant_job(jobs, got_job);
            JG_GLOBAL uint32_t* entry = as_global(job.fuse_ctl) + kFuseOwn + job.num_seq + got_i;
            while ((q = JG_FUSE_LOAD(entry)) == kFuseEmpty) {
                if (++polls > kFuseMaxPolls) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (threadIdx.x == 0) {
            if (q == kFuseEmpty) atomicAdd(&g_fuse_timeouts, 1u);
            s_claim[0] = got_job;
            s_claim[1] = q;
        }
    }
    __syncthreads(); // (what the parts stored is read with ld_shared below: no cache is asked to forget anything)
    const uint32_t seq = __builtin_amdgcn_readfirstlane(s_claim[1]);
    if (seq == kFuseEmpty) return;
    const ScanJob& job = constant_job(jobs, __builtin_amdgcn_readfirstlane(s_claim[0]));
    const JobView J(job);
    write_sequence<W, true>(J, static_cast<int>(seq), smem);
}
#undef JG_FUSE_LOAD
#undef JG_FUSE_ADD
#undef JG_FUSE_STORE

// ------------------------------------------------------------------------------------------------
// dequantisation + inverse DCT
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int unfixo(int x) { return (x + 0x1000) >> 13; }

/// 8-point fixed-point inverse DCT, the arithmetic of the reference's `idct_vector`
/// (src/idct.cu:49-95): Q15 even part, Q13 odd part, results rounded to int16.
/// kRound = 0x8000 is the reference's rounding; the row pass adds the level shift as well (128 in the high half:
/// the int16 the reference stores and then offsets, `(int16)(t + 128)`, src/idct.cu:218, wraps the same way).
template <int kRound = 0x8000>
__device__ __forceinline__ void idct8(int (&v)[8])
{
    constexpr int cos_1_4 = 0x5a82, sin_1_8 = 0x30fc, cos_1_8 = 0x7642;
    constexpr int osin_1_16 = 0x063e, osin_5_16 = 0x1a9b, ocos_1_16 = 0x1f63, ocos_5_16 = 0x11c7;

    const int e0 = (v[0] + v[4]) * cos_1_4;
    const int e1 = (v[0] - v[4]) * cos_1_4;
    // (a rotation as three products: x s - y c = (x + y) s - y (s + c), y s + x c = (x + y) s + x (c - s); the same
    // numbers in wrapping 32-bit arithmetic, and an addition issues faster than a multiplication)
    int z26 = (v[2] + v[6]) * sin_1_8;
    asm("" : "+v"(z26)); // one product with two users: left to itself the compiler multiplies it out again inside each of them
    const int e2  = z26 - v[6] * (sin_1_8 + cos_1_8);
    const int e3  = z26 + v[2] * (cos_1_8 - sin_1_8);
    const int a0 = e0 + e3, a1 = e1 + e2, a2 = e1 - e2, a3 = e0 - e3;

    const int m0 = unfixo((v[3] + v[5]) * cos_1_4);
    const int m1 = unfixo((v[3] - v[5]) * cos_1_4);
    // x4 written as a multiplication: `<<` on a negative int is undefined before C++20 and hipcc uses that
    const int q1 = v[1] * 4, q7 = v[7] * 4;
    const int o0 = q1 + m0, o1 = q7 + m1, o2 = q1 - m0, o3 = q7 - m1;
    int z01 = (o0 + o1) * osin_1_16;
    asm("" : "+v"(z01));
    const int b0  = z01 + o0 * (ocos_1_16 - osin_1_16); // o0 c + o1 s
    const int b1  = z01 - o1 * (ocos_1_16 + osin_1_16); // o0 s - o1 c
    int z23 = (o2 + o3) * osin_5_16;
    asm("" : "+v"(z23));
    const int b2  = z23 + o2 * (ocos_5_16 - osin_5_16); // o2 c + o3 s
    const int b3  = z23 - o3 * (ocos_5_16 + osin_5_16); // o2 s - o3 c

    // results rounded but NOT shifted: the int16 the reference stores (`unfixh`) is the high half
    v[0] = a0 + b0 + kRound;
    v[1] = a1 + b3 + kRound;
    v[2] = a2 + b2 + kRound;
    v[3] = a3 + b1 + kRound;
    v[4] = a3 - b1 + kRound;
    v[5] = a2 - b2 + kRound;
    v[6] = a1 - b3 + kRound;
    v[7] = a0 - b0 + kRound;
}

__device__ __forceinline__ uint32_t magic_quot(uint32_t n, uint32_t mul, uint32_t shift)
{
    return mul ? __umulhi(n, mul) >> shift : n;
}

/// Four finished samples from four row-pass results whose high halves already hold (int16)(t + 128): clamped to
/// 0..255 (reference src/idct.cu:218-220), one byte each.
__device__ __forceinline__ uint32_t finish_pixels(int w0, int w1, int w2, int w3)
{
    const uint32_t lo = __builtin_amdgcn_perm(static_cast<uint32_t>(w1), static_cast<uint32_t>(w0), 0x07060302u);
    const uint32_t hi = __builtin_amdgcn_perm(static_cast<uint32_t>(w3), static_cast<uint32_t>(w2), 0x07060302u);
    uint32_t a, b;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(a) : "v"(lo)); // two bytes in the low half
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b) : "v"(hi));
    return __builtin_amdgcn_perm(b, a, 0x05040100u);
}

constexpr int kIdctDuPerBlock = 32; // 8 lanes per data unit, 256 lanes
constexpr int kIdctDuStride   = 64 + 8; // int16 per staged data unit (+8: the 8 units of a wave start on different banks)

__device__ __forceinline__ void unpack8(const uint4& raw, int (&v)[8])
{
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = static_cast<int16_t>(w[i >> 1] >> (16 * (i & 1)));
}

constexpr int kIdctIters    = 8;                            // groups of 32 data units per workgroup
constexpr int kIdctDuPerWg  = kIdctDuPerBlock * kIdctIters; // 256

/// Zig-zag index of the coefficient in `row`, `col` (T.81 figure A.6), worked out instead of looked up: a table in
/// memory is one more load for the prologue to wait for. Diagonal d = row + col holds the indices from d (d + 1) / 2
/// on, downwards for odd d; the lower right half mirrors the upper left.
__host__ __device__ constexpr int zigzag_of(int row, int col)
{
    const bool low = row + col > 7;
    const int r = low ? 7 - row : row, c = low ? 7 - col : col, d = r + c;
    const int z = d * (d + 1) / 2 + ((d & 1) ? r : c);
    return low ? 63 - z : z;
}
constexpr bool zigzag_of_matches_table()
{
    constexpr uint8_t nat[64] = JG_ORDER_NATURAL; // zig-zag index -> natural index
    for (int z = 0; z < 64; ++z)
        if (zigzag_of(nat[z] >> 3, nat[z] & 7) != z) return false;
    return true;
}
static_assert(zigzag_of_matches_table(), "zigzag_of");

/// Two 16-bit products at once (v_pk_mul_lo_u16): the low halves of coefficient * quantiser.
__device__ __forceinline__ uint32_t mul_lo_u16x2(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, static_cast<u16x2>(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b)));
}

/// One data unit per 8 lanes, kIdctIters groups of 32 units per workgroup. The unit's entries are
/// gathered from the symbol stream (aligned 4-byte reads of two entries) and de-zigzagged on the
/// way into LDS; everything else is zero. The lane of the column pass dequantises its column when it
/// reads it. Steps and int16 truncation points are those of the reference `idct_kernel`
/// (src/idct.cu:146-223): (int16)(coef * q) -> column pass -> row pass -> +128 -> clamp. The MCU
/// geometry (reference decode_transpose.cu:65-131) is applied when the 8x8 pixels are stored.
///
/// Two things bound a naive version: LDS instruction issue and the chain of dependent loads
/// (table entry -> symbol entries) paid once per tiny workgroup. So the block is staged TRANSPOSED
/// ([column][row]: zeroing is one 16-byte write, the column pass one 16-byte read), all table
/// entries of the workgroup are loaded up front, and the first entries of each lane are fetched two
/// iterations ahead.
template <class JS>
__global__ __launch_bounds__(256) void idct_kernel(JS js)
{
    __shared__ __attribute__((aligned(16))) int16_t s_blk[kIdctDuPerBlock][kIdctDuStride]; // [unit][col * 8 + row]
    // [quantisation table][column][row]: the 16 bytes a lane of the column pass multiplies its column with
    // ((int16)(coef * q), reference idct.cu:178-180: the low 16 bits of the product, whatever the signs)
    __shared__ __attribute__((aligned(16))) uint16_t s_qcol[4 * 64];
    // zig-zag index -> byte offset of the coefficient's transposed slot in a staged block. 64 bytes are 16 banks:
    // lanes that ask for different entries never collide (same word: broadcast).
    __shared__ __attribute__((aligned(16))) uint8_t s_slot[64];
    __shared__ uint2 s_px[2][kIdctDuPerBlock][9]; // finished pixel rows, [buffer][unit][row] (+1: bank spread)
    // Where the pixels of each of the workgroup's data units go, worked out ONCE per unit by lane = unit (reference
    // decode_transpose.cu:65-131 walks the same geometry): address of the unit's top-left pixel, pitch, how many of
    // its 8 columns / rows are inside the plane (0..8), and its quantisation table. The per-iteration code reads
    // 16 bytes instead of redoing two divisions and a dozen multiply-adds per lane and unit row.
    struct UnitGeo {
        uint32_t addr_lo, addr_hi;
        int pitch;
        // rows to store (0..8; 0 if no column is visible) | byte offset of the quantisation table in s_qcol (bits
        // 7-8) | visible columns << 12 | kGeoWhole: every field where one instruction picks it up
        uint32_t vis;
    };
    constexpr uint32_t kGeoWhole = 1u << 31; // all 8 columns visible and every row 8-byte aligned: one store per row
    __shared__ __attribute__((aligned(16))) UnitGeo s_geo[kIdctDuPerWg];

    const JobView J(js.get());
    const IdctParams& ip = J.ip;
    const int du0        = blockIdx.x * kIdctDuPerWg;
    const int num_du     = ip.num_du;
    if (du0 >= num_du) return;

    const int t  = threadIdx.x;
    const int r  = t & 7;  // column (pass 1) or row (pass 2) handled by this lane
    const int dl = t >> 3; // data unit inside the group

    // The records of the lane's units of all iterations, asked for before anything else and without a branch (a unit
    // past the end reads the last record and counts no entries): eight loads in flight at once. Behind an `if` each,
    // as up to round 4, the compiler waited for every one before it issued the next -- eight memory latencies in a
    // row in front of the first iteration, in a workgroup that lives for eight iterations.
    uint2_t rec[kIdctIters];
#pragma unroll
    for (int it = 0; it < kIdctIters; ++it) rec[it] = ld_global(J.du_tab + min(du0 + it * kIdctDuPerBlock + dl, num_du - 1));
    // (and the lane's byte and word of the job's geometry tables, below)
    const uint32_t unit_byte = reinterpret_cast<const uint8_t*>(ip.du_comp)[t & 31];
    const uint32_t comp_word = reinterpret_cast<const uint32_t*>(ip.comp_h)[t & 31];

    s_qcol[t] = J.qtables[(t & ~63) + (t & 7) * 8 + ((t >> 3) & 7)]; // [table][col][row] <- natural row * 8 + col
    if (t < 64) s_slot[zigzag_of(t >> 3, t & 7)] = static_cast<uint8_t>(((t & 7) * 8 + (t >> 3)) * 2); // natural row * 8 + col -> transposed slot col * 8 + row
    // Geometry, first half: which unit of which MCU, and the one load the rest depends on. Straight-line code (a unit
    // past the end works on the last one and is marked invisible): the loads of this prologue are then the
    // compiler's to count, and the first entries below travel while the geometry is worked out.
    const int gdu  = min(du0 + t, num_du - 1);
    const int grel = static_cast<int>(magic_quot(gdu, ip.du_per_mcu_mul, ip.du_per_mcu_shift));
    const int gk   = gdu - grel * ip.du_per_mcu;
    // What the unit's place depends on sits in two small tables of the job: the MCU's units (component, block column,
    // block row: three arrays of 10 bytes) and the components (six arrays of 4 ints, then 4 plane pointers). Lane L of
    // every half wave loads byte L of the first and word L of the second -- loads that depend on nothing -- and a unit
    // then takes its values from the lanes that hold them (ds_bpermute: no memory behind it). Indexed loads, first
    // by the unit's place in the MCU, then by its component, were two more memory latencies in a row.
    static_assert(kMaxDuPerMcu == 10 && kMaxComp == 4 && sizeof(ip.plane[0]) == 8, "lane layout of the two tables");
    static_assert(offsetof(IdctParams, du_dx) == offsetof(IdctParams, du_comp) + 10 && offsetof(IdctParams, du_dy) == offsetof(IdctParams, du_comp) + 20 &&
                      offsetof(IdctParams, comp_h) >= offsetof(IdctParams, du_comp) + 32,
                  "32 bytes from du_comp on");
    static_assert(offsetof(IdctParams, comp_v) == offsetof(IdctParams, comp_h) + 16 && offsetof(IdctParams, size_x) == offsetof(IdctParams, comp_h) + 32 &&
                      offsetof(IdctParams, size_y) == offsetof(IdctParams, comp_h) + 48 && offsetof(IdctParams, pitch) == offsetof(IdctParams, comp_h) + 64 &&
                      offsetof(IdctParams, qidx) == offsetof(IdctParams, comp_h) + 80 && offsetof(IdctParams, plane) == offsetof(IdctParams, comp_h) + 96,
                  "32 words from comp_h on");
    const auto from_lane = [](int lane, uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(lane * 4, static_cast<int>(v))); };
    const int gmcu = grel + ip.first_mcu;
    const int gmy  = static_cast<int>(magic_quot(gmcu, ip.mcus_x_mul, ip.mcus_x_shift));
    const int gmx  = gmcu - gmy * ip.mcus_x;
    // a table entry that was never written (corrupt stream) must not lead out of the buffer
    uint32_t toff[kIdctIters], tcnt[kIdctIters];
    const uint64_t limit = J.sym_entries - 10 * kSymSectorStride; // a 128-entry gather from here stays inside
#pragma unroll
    for (int it = 0; it < kIdctIters; ++it) {
        const bool mine = du0 + it * kIdctDuPerBlock + dl < num_du;
        tcnt[it] = mine ? rec[it].y & 0xFFu : 0u; // entries (at most 127) | kUnitHasEscape
        toff[it] = static_cast<uint32_t>(rec[it].x < limit ? rec[it].x : limit);
    }
    // Nothing but the fetched words to place in any of this wave's units (no unit above 31 entries, none with an
    // escape: the record's flag sits above the count)? Asked once per wave, not once per iteration.
    uint32_t most = 0;
#pragma unroll
    for (int it = 0; it < kIdctIters; ++it) most = max(most, tcnt[it]);
    const bool plain = __ballot(most > 31u) == 0;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 zero4 = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(zero4)); // four registers that stay zero: the compiler would set them up again in every iteration
    const uint2* const px_mine = &s_px[0][t & 31][t >> 5];
    const uint8_t* const qcol_mine = reinterpret_cast<const uint8_t*>(s_qcol) + r * 16;
#ifndef JG_IDCT_PAIRS
#define JG_IDCT_PAIRS 2
#endif
#ifndef JG_IDCT_DEPTH
#define JG_IDCT_DEPTH 2
#endif
#ifndef JG_IDCT_UNCOND
#define JG_IDCT_UNCOND 1
#endif
    // Entry PAIRS per lane, fetched kDepth iterations ahead: the lane reads the aligned 32-bit words r, r + 8, ... of
    // the sector row its unit starts in, counted from the word that holds the unit's first entry. Eight words further
    // is the same word of the next sector (one 32-byte sector = 16 entries = 8 words): +2048 bytes. With an odd first
    // entry the low half of lane 0's first word belongs to the unit in front. The words are read whether the unit
    // reaches them or not (toff is clamped so that they lie inside the buffer; what is not the unit's is not placed):
    // without branches around the loads the compiler can count them, and the wait for one iteration's words leaves
    // the next one's in flight. (Up to round 4 the lanes read single entries under a compare and a branch each: twice
    // the loads, twice the address arithmetic, and one wait for everything.)
    constexpr int kPairs = JG_IDCT_PAIRS;
    static_assert(kPairs * kSymSectorStride * 2 <= 4096 + 2048, "immediate offsets of the loads");
    constexpr int kDepth = JG_IDCT_DEPTH; // iterations the fetches run ahead
    uint32_t pre[kIdctIters + kDepth][kPairs];
    const auto entry_at = [&](uint32_t index) -> uint32_t {
        return *reinterpret_cast<JG_GLOBAL const uint16_t*>(reinterpret_cast<JG_GLOBAL const uint8_t*>(J.sym) + index * 2u);
    };
    // Entry j of the unit sits in half (j + odd) & 1 of word (j + odd) / 2 counted as above; a lane's word k holds
    // the entries jb + 16 k and jb + 16 k + 1, jb = 2 r - odd.
    const auto prefetch = [&](uint32_t first, uint32_t cnt, uint32_t (&out)[kPairs]) {
        // the word that holds the unit's first entry, r words on; past the end of the 8-word sector: the next sector
        const uint32_t word = first >> 1;
        const uint32_t over = ((word & 7u) + static_cast<uint32_t>(r)) & 8u;
        const uint32_t base = word * 4u + static_cast<uint32_t>(r) * 4u + over * ((kSymSectorStride * 2u - 32u) / 8u);
        const int jb        = 2 * r - static_cast<int>(first & 1u);
        JG_GLOBAL const uint8_t* stream = reinterpret_cast<JG_GLOBAL const uint8_t*>(J.sym);
#pragma unroll
        for (int k = 0; k < kPairs; ++k)
            out[k] = (JG_IDCT_UNCOND || jb + 16 * k < static_cast<int>(cnt)) ? *reinterpret_cast<JG_GLOBAL const uint32_t*>(stream + (base + k * (kSymSectorStride * 2u))) : 0u;
    };

    {
        // (the first entries: asked for behind the geometry's loads, so that waiting for those does not wait for these)
#pragma unroll
        for (int d = 0; d < kDepth; ++d) prefetch(toff[d], tcnt[d] & 0x7Fu, pre[d]);
        const int sc = static_cast<int>(from_lane(gk, unit_byte));
        const int dx = static_cast<int>(from_lane(10 + gk, unit_byte)), dy = static_cast<int>(from_lane(20 + gk, unit_byte));
        const int comp_h = static_cast<int>(from_lane(sc, comp_word)), comp_v = static_cast<int>(from_lane(4 + sc, comp_word));
        const int size_x = static_cast<int>(from_lane(8 + sc, comp_word)), size_y = static_cast<int>(from_lane(12 + sc, comp_word));
        const int pitch = static_cast<int>(from_lane(16 + sc, comp_word)), qidx = static_cast<int>(from_lane(20 + sc, comp_word));
        const uint64_t plane = static_cast<uint64_t>(from_lane(24 + 2 * sc, comp_word)) | static_cast<uint64_t>(from_lane(25 + 2 * sc, comp_word)) << 32;
        const int x0  = (gmx * comp_h + dx) * 8;
        const int y0  = (gmy * comp_v + dy) * 8;
        const int vx  = min(max(size_x - x0, 0), 8), vy = min(max(size_y - y0, 0), 8);
        const uint64_t a = plane + static_cast<uint64_t>(y0) * static_cast<uint32_t>(pitch) + static_cast<uint32_t>(x0);
        const bool whole = vx == 8 && ((a | static_cast<uint32_t>(pitch)) & 7u) == 0;
        const uint32_t vis = static_cast<uint32_t>(vx > 0 ? vy : 0) | static_cast<uint32_t>(qidx & 3) << 7 | static_cast<uint32_t>(vx) << 12 | (whole ? kGeoWhole : 0u);
        s_geo[t] = UnitGeo{static_cast<uint32_t>(a), static_cast<uint32_t>(a >> 32), pitch, du0 + t < num_du ? vis : 0u};
    }
    int16_t* blk = s_blk[dl];
    uint8_t* const blk_bytes = reinterpret_cast<uint8_t*>(blk);
    __syncthreads(); // s_qcol, s_slot, s_geo are loaded

#pragma unroll
    for (int it = 0; it < kIdctIters; ++it) {
        uint32_t ex[kPairs];
#pragma unroll
        for (int k = 0; k < kPairs; ++k) ex[k] = pre[it][k];
        if (it + kDepth < kIdctIters) prefetch(toff[it + kDepth], tcnt[it + kDepth] & 0x7Fu, pre[it + kDepth]); // in flight while this one computes
        // The 8 lanes of a data unit sit in one wave and LDS executes a wave's instructions in order,
        // so the phases below need no workgroup barrier among themselves; only the pixel re-mapping
        // at the end crosses waves (one barrier per iteration, buffers alternate).
        *reinterpret_cast<u32x4*>(blk + r * 8) = zero4;

        const uint32_t qoff = s_geo[it * kIdctDuPerBlock + dl].vis & 0x180u;
        // place one coefficient, not yet dequantised: zig-zag index, value (its low 16 bits count)
        const auto put = [&](uint32_t zz, uint32_t value) { *reinterpret_cast<int16_t*>(blk_bytes + s_slot[zz]) = static_cast<int16_t>(value); };
        const uint32_t cnt = tcnt[it] & 0x7Fu;
        const bool odd     = (toff[it] & 1u) != 0;
        // Entry j of the unit (jg_defs.h): j == 0 is the DC value; an AC entry holds value << 6 | index; an entry
        // with index 0 behind one is the ESCAPE that carries the value's high bits. The unit's record says whether it
        // holds one (no photograph does).
        if (__builtin_expect(plain || __ballot((tcnt[it] & kUnitHasEscape) != 0) == 0, 1)) {
            // the look-ups first, all of them (an index of a word that was not loaded is 0): one LDS latency, not one per entry
            uint32_t slot[kPairs][2];
#pragma unroll
            for (int k = 0; k < kPairs; ++k) {
                slot[k][0] = s_slot[ex[k] & 63u];
                slot[k][1] = s_slot[(ex[k] >> 16) & 63u];
            }
            if (r == 0) blk[0] = static_cast<int16_t>(ex[0] >> ((toff[it] << 4) & 31u)); // DC: lane 0, the half the unit starts in
            // The lane's words hold the entries jb + c, c = 16 k + h, jb = 2 r - odd; an AC entry of the unit is one
            // with 1 <= jb + c < cnt: c < left, and for lane 0 not the DC or the entry in front of it.
            const int left = static_cast<int>(cnt) + static_cast<int>(toff[it] & 1u) - 2 * r;
#pragma unroll
            for (int k = 0; k < kPairs; ++k) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bool mine = 16 * k + h < left;
                    if (k == 0 && h == 0) mine = mine && r != 0;
                    if (k == 0 && h == 1) mine = mine && !(r == 0 && odd);
                    if (mine)
                        *reinterpret_cast<int16_t*>(blk_bytes + slot[k][h]) = static_cast<int16_t>(static_cast<int32_t>(ex[k] << (16 - 16 * h)) >> 22);
                }
            }
            if (!plain) {
                for (uint32_t i = 16u * kPairs - (toff[it] & 1u) + static_cast<uint32_t>(r); i < cnt; i += 8) { // dense units only: entries behind the fetched words
                    const uint32_t e = entry_at(sym_advance(toff[it], i));
                    put(sym_entry_index(e), static_cast<uint32_t>(sym_entry_value(e)));
                }
            }
        } else {
            for (uint32_t i = r; i < cnt; i += 8) {
                const uint32_t e    = entry_at(sym_advance(toff[it], i));
                const uint32_t next = i + 1 < cnt ? entry_at(sym_advance(toff[it], i + 1)) : 1u;
                if (i == 0) put(0, e);
                else if (sym_entry_index(e) != 0)
                    put(sym_entry_index(e), static_cast<uint32_t>(sym_entry_index(next) == 0 ? sym_entry_value(e, next) : sym_entry_value(e)));
                asm volatile("" ::"v"(next)); // no load of this rare path is left in flight: the common path behind it would wait for it with everything else
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int v[8];
        {
            uint4 col      = *reinterpret_cast<const uint4*>(blk + r * 8); // column r
            const uint4 qc = *reinterpret_cast<const uint4*>(qcol_mine + qoff);
            col.x = mul_lo_u16x2(col.x, qc.x), col.y = mul_lo_u16x2(col.y, qc.y), col.z = mul_lo_u16x2(col.z, qc.z), col.w = mul_lo_u16x2(col.w, qc.w);
            unpack8(col, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // every column is read before rows overwrite the block
        idct8(v);
#pragma unroll
        for (int i = 0; i < 8; ++i) blk[i * 8 + r] = static_cast<int16_t>(v[i] >> 16); // now [row][col]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        unpack8(*reinterpret_cast<const uint4*>(blk + r * 8), v); // row r
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // row reads precede the next iteration's zeroing
        idct8<0x8000 + (128 << 16)>(v);

        uint2 o;
        o.x = finish_pixels(v[0], v[1], v[2], v[3]);
        o.y = finish_pixels(v[4], v[5], v[6], v[7]);
        // A lane holds row r of unit dl; storing that directly makes every wave store touch ~40 cache
        // lines (8 units x 8 rows). Re-map through LDS: lane -> (row t / 32, unit t % 32), so that
        // consecutive lanes write the neighbouring 8-byte segments of one image row.
        s_px[it & 1][dl][r] = o;
        __syncthreads();
        // The next iteration's entries have had an iteration's time or more to arrive; asking for them HERE, in front
        // of the pixel stores, keeps those stores out of the wait (one counter counts loads and stores, and behind the
        // stores' branches the compiler can only wait for everything: with the wait at the first use every iteration
        // stood until its predecessor's pixels had reached L2 and its own entries had arrived, fetched a placement
        // phase earlier).
        if (it + 1 < kIdctIters) {
#pragma unroll
            for (int k = 0; k < kPairs; ++k) asm volatile("" : "+v"(pre[it + 1][k]));
        }
        {
            const int r2      = t >> 5;
            const int j       = t & 31;
            const UnitGeo g   = s_geo[it * kIdctDuPerBlock + j]; // all zero behind the last unit: nothing visible
            const int vx = (g.vis >> 12) & 15;
            if (r2 < static_cast<int>(g.vis & 15u)) {
                const uint2 w = px_mine[(it & 1) * (kIdctDuPerBlock * 9)];
                JG_GLOBAL uint8_t* row = reinterpret_cast<JG_GLOBAL uint8_t*>(
                    ((static_cast<uint64_t>(g.addr_hi) << 32) | g.addr_lo) + static_cast<uint64_t>(static_cast<uint32_t>(r2)) * static_cast<uint64_t>(static_cast<uint32_t>(g.pitch))); // one v_mad_u64_u32
                if (g.vis & kGeoWhole) {
                    st_global(reinterpret_cast<JG_GLOBAL uint2*>(row), w);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i < vx) row[i] = static_cast<uint8_t>((i < 4 ? w.x >> (8 * i) : w.y >> (8 * (i - 4))) & 0xFFu);
                    }
                }
            }
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

/// Chroma replication + YCbCr -> RGB, interleaved 8-bit output; the arithmetic of the reference's host
/// helper `conv_to_rgbi` (util/util.h:62-104): nearest-neighbour replication, the JFIF matrix in float,
/// roundf, clamp. Each lane produces 4 consecutive pixels (12 bytes) of one row, so a wave writes 768
/// contiguous bytes. With one component the sample is copied to R, G and B (util.h:47-58).
struct RgbiParams {
    const uint8_t* plane[3];
    int pitch[3], w[3], h[3];
    int num_x[3], num_y[3]; // sampling factors; the maxima are the denominators
    int den_x, den_y;
    int ncomp;
};

__global__ __launch_bounds__(256) void rgbi_kernel(RgbiParams p, uint8_t* __restrict__ dst, int dst_pitch, int width, int height)
{
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y  = blockIdx.y;
    if (x0 >= width || y >= height) return;
    uint32_t out[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = min(x0 + i, width - 1);
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cc = c < p.ncomp ? c : 0;
            const int sy = min(y * p.num_y[cc] / p.den_y, p.h[cc] - 1);
            const int sx = min(x * p.num_x[cc] / p.den_x, p.w[cc] - 1);
            v[c]         = static_cast<float>(p.plane[cc][static_cast<size_t>(sy) * p.pitch[cc] + sx]);
        }
        float r = v[0], g = v[0], b = v[0];
        if (p.ncomp == 3) {
            r = v[0] + 1.402f * (v[2] - 128.f);
            g = v[0] - .344136f * (v[1] - 128.f) - .714136f * (v[2] - 128.f);
            b = v[0] + 1.772f * (v[1] - 128.f);
        }
        out[3 * i + 0] = static_cast<uint32_t>(fmaxf(0.f, fminf(roundf(r), 255.f)));
        out[3 * i + 1] = static_cast<uint32_t>(fmaxf(0.f, fminf(roundf(g), 255.f)));
        out[3 * i + 2] = static_cast<uint32_t>(fmaxf(0.f, fminf(roundf(b), 255.f)));
    }
    uint8_t* drow = dst + static_cast<size_t>(y) * dst_pitch + static_cast<size_t>(x0) * 3;
    if (x0 + 4 <= width && (reinterpret_cast<uintptr_t>(drow) & 3) == 0) {
        uint32_t* d = reinterpret_cast<uint32_t*>(drow);
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = out[4 * k] | out[4 * k + 1] << 8 | out[4 * k + 2] << 16 | out[4 * k + 3] << 24;
    } else {
        for (int i = 0; i < 12 && x0 + i / 3 < width; ++i) drow[i] = static_cast<uint8_t>(out[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------

template <class K>
hipError_t allow_lds(K kernel, size_t bytes)
{
    // more than 64 KiB of dynamic LDS has to be requested per kernel; gfx950 has 160 KiB per CU
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

/// Does a batch launch with these extents run huff_tail_write? Asked by the three stages it replaces or skips: only behind
/// huff_sync_intra_batch (which clears its flags) and only where the tail kernel would run its 256-lane workgroups.
inline bool fuses_tail_write(const JobExtent& e, int num_jobs)
{
    return e.fuse_tail_write && !e.repack_flows && e.max_tail_parts > 0 && e.max_tail_part < kTailLargeFrom && e.max_seq > 0 && num_jobs <= kFuseMaxJobs;
}

template <int W, class JS>
hipError_t launch_huff(Stage stage, const JS& js, const JobExtent& e, int grid_y, hipStream_t stream)
{
    const size_t seq_lds = SeqLds::kTabs + e.max_tab_bytes_sync;
    hipError_t err       = hipSuccess;
    // the cursor ring holds absolute LDS addresses in 16 bits (load_tables): refuse rather than wrap
    if (e.max_tab_bytes > kMaxTablePack || e.max_tab_bytes_sync > kMaxTablePackSync) return hipErrorInvalidValue;
    switch (stage) {
    case kStageSyncIntra:
        if constexpr (JS::kRepackFlows) {
            if ((err = allow_lds(huff_sync_intra<W, JS>, seq_lds)) != hipSuccess) return err;
            huff_sync_intra<W, JS><<<dim3(e.max_seq, grid_y), T, seq_lds, stream>>>(js);
        } else if (e.repack_flows) { // a batch too small to fill the chip: the lone decode's sequence kernel (JobArrayLow)
            if ((err = allow_lds(huff_sync_intra<W, JobArrayLow>, seq_lds)) != hipSuccess) return err;
            huff_sync_intra<W, JobArrayLow><<<dim3(e.max_seq, grid_y), T, seq_lds, stream>>>(JobArrayLow{js.jobs});
        } else {
            const size_t lds = SeqLdsBatch::kTabs + e.max_tab_bytes_sync;
            if ((err = allow_lds(huff_sync_intra_batch<W, JS>, lds)) != hipSuccess) return err;
            huff_sync_intra_batch<W, JS><<<dim3((e.max_seq + kBatchSeqPerWg - 1) / kBatchSeqPerWg, grid_y), T * kBatchSeqPerWg, lds, stream>>>(js);
        }
        break;
    case kStageSyncInter:
        if (e.max_tail_parts > 0 && !(std::is_same<JS, JobArray>::value && fuses_tail_write(e, grid_y)))
        {
            if (e.max_tail_part >= kTailLargeFrom) {
                constexpr int TL = kTailLanesLarge;
                huff_sync_tail<W, TL, JS><<<dim3(e.max_tail_parts, grid_y), TL, 3 * TL * 4 + e.max_tab_bytes_sync, stream>>>(js);
            } else {
                constexpr int TL = kTailLanesSmall;
                huff_sync_tail<W, TL, JS><<<dim3(e.max_tail_parts, grid_y), TL, 3 * TL * 4 + e.max_tab_bytes_sync, stream>>>(js);
            }
        }
        break;
    case kStageWrite: {
        size_t lds = WriteLds::kTabs + e.max_tab_bytes;
#if defined(JG_PROBE) // occupancy experiments: JPEGGPU_EXP_EXTRA_LDS bytes of LDS nobody uses (fewer workgroups per CU)
        if (const char* x = std::getenv("JPEGGPU_EXP_EXTRA_LDS")) lds += static_cast<size_t>(std::atoi(x));
        if ((err = allow_lds(huff_write<W, JS>, lds)) != hipSuccess) return err;
#endif
        if constexpr (std::is_same<JS, JobArray>::value) {
            if (fuses_tail_write(e, grid_y)) { // the parts of the tail kernel and the sequences of the write pass, by ticket
                const size_t tail_lds = 3 * T * 4 + e.max_tab_bytes_sync;
                lds                   = lds > tail_lds ? lds : tail_lds;
                if ((err = allow_lds(huff_tail_write<W>, lds)) != hipSuccess) return err;
                const long long wgs = static_cast<long long>(grid_y) * (static_cast<long long>(e.max_tail_parts) + e.max_seq);
                if (wgs > 0x7FFFFFFFll) return hipErrorInvalidValue;
                huff_tail_write<W><<<dim3(static_cast<unsigned>(wgs)), T, lds, stream>>>(js.jobs, grid_y, e.max_tail_parts, e.max_seq);
                break;
            }
        }
        huff_write<W, JS><<<dim3(e.max_seq, grid_y), T, lds, stream>>>(js);
        break;
    }
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <class JS>
hipError_t launch_any(Stage stage, const JS& js, const JobExtent& e, int grid_y, hipStream_t stream)
{
    switch (stage) {
    case kStageFront:
        return hipSuccess; // nothing to clear any more: the write pass emits a symbol stream
    case kStageDestuff:
        if (e.max_chunks == 0) return hipSuccess;
        destuff_kernel<JS><<<dim3((e.max_chunks + kDestuffChunksPerWg - 1) / kDestuffChunksPerWg, grid_y), 256, 0, stream>>>(js);
        return hipGetLastError();
    case kStageTails:
        // (folded into the tail kernel's 256-lane workgroups: part_seq_tails; parts of scans without restart markers run the
        // 1024-lane kernel and keep this launch)
        if (e.max_seq == 0 || (e.max_tail_parts > 0 && e.max_tail_part < kTailLargeFrom)) return hipSuccess;
        huff_seq_tails<JS><<<dim3(e.max_seq, grid_y), T, 0, stream>>>(js);
        return hipGetLastError();
    case kStageIdct:
        if (e.max_idct_blocks == 0) return hipSuccess;
        idct_kernel<JS><<<dim3(e.max_idct_blocks, grid_y), 256, 0, stream>>>(js);
        return hipGetLastError();
    case kStageSyncIntra:
    case kStageSyncInter:
    case kStageWrite:
        if (e.max_seq == 0) return hipSuccess;
        switch (e.subseq_words) {
        case 8: return launch_huff<8, JS>(stage, js, e, grid_y, stream);
        case 16: return launch_huff<16, JS>(stage, js, e, grid_y, stream);
        case 32: return launch_huff<32, JS>(stage, js, e, grid_y, stream);
        case 64: return launch_huff<64, JS>(stage, js, e, grid_y, stream);
        }
        return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
}

} // namespace

#if defined(JG_PROBE)
extern "C" __attribute__((visibility("default"))) int jpeggpu_probe_read_write(unsigned long long* dst4, int clear)
{
    if (hipMemcpyFromSymbol(dst4, HIP_SYMBOL(g_probe_write), sizeof(g_probe_write)) != hipSuccess) return 1;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_probe_write)) != hipSuccess || hipMemset(p, 0, sizeof(g_probe_write)) != hipSuccess) return 2;
    }
    return 0;
}

extern "C" __attribute__((visibility("default"))) int jpeggpu_probe_read_lane_iters(uint16_t* dst, size_t count)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_probe_lane_iters), count * 2) == hipSuccess ? 0 : 1;
}

extern "C" __attribute__((visibility("default"))) int jpeggpu_probe_read_lane_rare(uint32_t* dst, size_t count)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_probe_lane_rare), count * 4) == hipSuccess ? 0 : 1;
}

extern "C" __attribute__((visibility("default"))) int jpeggpu_probe_read_tail(uint32_t* dst, size_t count)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_probe_tail), count * 4) == hipSuccess ? 0 : 1;
}

extern "C" __attribute__((visibility("default"))) int jpeggpu_probe_read(void* dst, size_t bytes, int clear)
{
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_probe), bytes) != hipSuccess) return 1;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_probe)) != hipSuccess || hipMemset(p, 0, sizeof(g_probe)) != hipSuccess) return 2;
    }
    return 0;
}
#endif

hipError_t read_fuse_timeouts(unsigned int* count)
{
    return hipMemcpyFromSymbol(count, HIP_SYMBOL(g_fuse_timeouts), sizeof(unsigned int));
}

bool subseq_bytes_supported(int b) { return b == 32 || b == 64 || b == 128 || b == 256; }

void extend(JobExtent& e, const ScanJob& job)
{
    e.max_chunks      = job.num_chunks > e.max_chunks ? job.num_chunks : e.max_chunks;
    e.max_seq         = job.num_seq > e.max_seq ? job.num_seq : e.max_seq;
    const int blocks  = (job.ip.num_du + kIdctDuPerWg - 1) / kIdctDuPerWg;
    e.max_idct_blocks = blocks > e.max_idct_blocks ? blocks : e.max_idct_blocks;
    e.max_tab_bytes   = job.sp.tab_bytes > e.max_tab_bytes ? job.sp.tab_bytes : e.max_tab_bytes;
    e.max_tab_bytes_sync = job.sp.tab_bytes_sync > e.max_tab_bytes_sync ? job.sp.tab_bytes_sync : e.max_tab_bytes_sync;
    e.subseq_words    = job.sp.subseq_words;
    e.max_tail_parts  = job.num_tail_parts > e.max_tail_parts ? job.num_tail_parts : e.max_tail_parts;
    e.max_tail_part   = job.max_tail_part > e.max_tail_part ? job.max_tail_part : e.max_tail_part;
}

template <int W, class JS>
hipError_t launch_mh_w(const JS& js, const ScanJob& job, int max_seg_subseq, hipStream_t stream)
{
    const dim3 grid((job.sp.num_subseq + T - 1) / T, job.sp.mh);
    const size_t lds = job.sp.tab_bytes_sync;
    hipError_t err;
    if ((err = allow_lds(huff_mh_spec<W, JS>, lds)) != hipSuccess) return err;
    if ((err = allow_lds(huff_mh_flow<W, JS>, lds)) != hipSuccess) return err;
    huff_mh_spec<W, JS><<<grid, T, lds, stream>>>(js);
    huff_mh_flow<W, JS><<<grid, T, lds, stream>>>(js);
    const size_t rlds = mh_resolve_lds(job.sp.mh, max_seg_subseq);
    if (job.num_mh_blocks > 0) { // segments longer than the chain walk's LDS: block by block (jg_defs.h)
        const size_t mlds = static_cast<size_t>(job.sp.mh) * kMhMaxSegSubseq * 4, clds = static_cast<size_t>(job.num_mh_blocks) * (128 + 4);
        if ((err = allow_lds(huff_mh_block_maps<JS>, mlds)) != hipSuccess) return err;
        if ((err = allow_lds(huff_mh_resolve<JS, true>, rlds)) != hipSuccess) return err;
        huff_mh_block_maps<JS><<<job.num_mh_blocks, 256, mlds, stream>>>(js);
        huff_mh_block_chain<JS><<<1, 256, clds, stream>>>(js);
        huff_mh_resolve<JS, true><<<job.num_mh_blocks, 256, rlds, stream>>>(js);
        return hipGetLastError();
    }
    if ((err = allow_lds(huff_mh_resolve<JS, false>, rlds)) != hipSuccess) return err;
    huff_mh_resolve<JS, false><<<job.sp.num_segments, 256, rlds, stream>>>(js);
    return hipGetLastError();
}

template <class JS>
hipError_t launch_mh_any(const JS& js, const ScanJob& job, int max_seg_subseq, hipStream_t stream)
{
    if (job.sp.num_subseq == 0 || job.sp.num_segments == 0) return hipSuccess; // an empty share of the segments: nothing to speculate
    if (job.sp.mh < 2 || job.sp.mh > kMhMaxHyp || max_seg_subseq < 1 || max_seg_subseq > kMhMaxSegSubseq) return hipErrorInvalidValue;
    if (job.sp.tab_bytes_sync > kMaxTablePackSync) return hipErrorInvalidValue;
    switch (job.sp.subseq_words) {
    case 8: return launch_mh_w<8>(js, job, max_seg_subseq, stream);
    case 16: return launch_mh_w<16>(js, job, max_seg_subseq, stream);
    case 32: return launch_mh_w<32>(js, job, max_seg_subseq, stream);
    case 64: return launch_mh_w<64>(js, job, max_seg_subseq, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mh(const ScanJob& job, const ScanJob* d_job, int max_seg_subseq, hipStream_t stream)
{
    // `job`: the host's copy (sizes: for a device-scanned image the capacities from the header); d_job: where the kernels
    // read the job from if the device has filled in its counts (jg_front.hip), else null: passed by value
    if (d_job) return launch_mh_any(JobSingle{d_job}, job, max_seg_subseq, stream);
    return launch_mh_any(JobByValue{job}, job, max_seg_subseq, stream);
}

hipError_t launch_stage(Stage stage, const ScanJob& job, hipStream_t stream)
{
    JobExtent e;
    extend(e, job);
    return launch_any(stage, JobByValue{job}, e, 1, stream);
}

hipError_t launch_stage_scans(Stage stage, const ScanJob* jobs, int num_jobs, hipStream_t stream)
{
    if (num_jobs < 1 || num_jobs > kMaxScans) return hipErrorInvalidValue;
    if (num_jobs == 1) return launch_stage(stage, jobs[0], stream);
    JobsByValue js{};
    JobExtent e;
    for (int i = 0; i < num_jobs; ++i) {
        js.jobs[i] = jobs[i];
        extend(e, jobs[i]);
    }
    return launch_any(stage, js, e, num_jobs, stream);
}

hipError_t launch_stage_device_job(Stage stage, const ScanJob* d_job, const JobExtent& extent, hipStream_t stream)
{
    return launch_any(stage, JobSingle{d_job}, extent, 1, stream);
}

hipError_t launch_stage_batch(
    Stage stage, const ScanJob* d_jobs, int num_jobs, const JobExtent& extent, hipStream_t stream)
{
    if (num_jobs <= 0) return hipSuccess;
    return launch_any(stage, JobArray{d_jobs}, extent, num_jobs, stream);
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

hipError_t launch_rgbi(
    const uint8_t* const* planes, const int* pitch, const int* w, const int* h, const int* num_x, const int* num_y,
    int den_x, int den_y, int ncomp, uint8_t* dst, int dst_pitch, int width, int height, hipStream_t stream)
{
    if (width <= 0 || height <= 0) return hipSuccess;
    RgbiParams p{};
    for (int c = 0; c < 3; ++c) {
        const int cc = c < ncomp ? c : 0;
        p.plane[c] = planes[cc];
        p.pitch[c] = pitch[cc];
        p.w[c]     = w[cc];
        p.h[c]     = h[cc];
        p.num_x[c] = num_x[cc];
        p.num_y[c] = num_y[cc];
    }
    p.den_x = den_x;
    p.den_y = den_y;
    p.ncomp = ncomp;
    const dim3 grid((width + 1023) / 1024, height);
    rgbi_kernel<<<grid, 256, 0, stream>>>(p, dst, dst_pitch, width, height);
    return hipGetLastError();
}

} // namespace jg
