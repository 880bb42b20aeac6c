// jg_front.hip -- device-side front end: finds the restart markers and the end of the
// entropy-coded segment on the GPU and builds, in device memory, what the host walk of jg_reader.cpp builds
// for the default path (reference src/reader.cpp:447-489): the segment table, the destuff work list with
// destination offsets, the tail-kernel parts and the counts the Huffman kernels read from their job.
//
//   front_count     one workgroup per 4 KiB window: data bytes (the byte rule of decode_destuff.cu:37-44) and markers
//   front_prefix    one workgroup: prefix sums over the windows
//   front_marks     per window again: every marker now knows its ordinal in the scan and the data bytes in
//                   front of it; the first expect_segments + 1 are recorded, the terminating one is found
//   front_plan      one workgroup: validation, prefix sums over segments, table building
//
// Only single-scan files take this path (everything behind the first scan's data would otherwise have to be
// parsed by the host anyway); sizes are upper bounds computed from the header.
#include "jg_front.hpp"

#include <hip/hip_runtime.h>

#include "jg_bytes.h"
#include "jg_wave.h"

namespace jg {

namespace {

#define JG_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ JG_GLOBAL T* as_global(T* p)
{
    return (JG_GLOBAL T*)p;
}

/// Exclusive prefix of `v` over the workgroup's NW waves; `total` gets the workgroup sum.
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave, uint32_t& total)
{
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads(); // previous use of s_wave is over
    if (lane_id() == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    total        = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t w = s_wave[k];
        off += k < static_cast<int>(threadIdx.x >> 6) ? w : 0u;
        total += w;
    }
    return off + incl - v;
}

/// Where a workgroup finds its parameters: one scan by value (the drop-in decode), or entry blockIdx.y of an
/// array in device memory (a batch).
struct FrontOne {
    static constexpr bool kCarriesJob = false;
    FrontParams P;
    __device__ __forceinline__ const FrontParams& get() const { return P; }
};
/// front_count of a single scan also carries the scan's job and stores it to device memory (jg_front.hpp).
struct FrontOneJob {
    static constexpr bool kCarriesJob = true;
    FrontParams P;
    ScanJob job;
    __device__ __forceinline__ const FrontParams& get() const { return P; }
};
struct FrontMany {
    static constexpr bool kCarriesJob = false;
    const FrontParams* arr;
    __device__ __forceinline__ FrontParams get() const { return arr[blockIdx.y]; }
};

/// The 16 bytes of a lane, classified four at a time (jg_bytes.h, masks with one bit per byte at bit 7): which
/// are data (byte rule of decode_destuff.cu:37-44) and which are the FF right in front of a marker code (followed
/// neither by a stuffed zero nor by a fill byte). Bytes in front of the scan read as 00 -- the byte in front of
/// the first entropy-coded byte belongs to the scan header, never stuffing, as in the host walk -- and bytes
/// behind the buffer as FF (never data, never a marker: its code would lie outside); `lead` counts the former,
/// which the rule takes for data.
struct LaneBytes {
    uint32_t word[4];
    uint32_t next;
    uint32_t data[4], mark[4]; // 0x80 domain
    uint32_t bad[4];           // an FF behind an FF and in front of a 00: "FF FF 00" is neither fill + marker nor stuffing
    uint32_t lead;
    __device__ __forceinline__ uint32_t data_count() const
    {
        return __popc(data[0]) + __popc(data[1]) + __popc(data[2]) + __popc(data[3]) - lead;
    }
    __device__ __forceinline__ uint32_t mark_count() const { return __popc(mark[0]) + __popc(mark[1]) + __popc(mark[2]) + __popc(mark[3]); }
    /// One bit per byte, bit i = byte i.
    __device__ __forceinline__ uint32_t data_mask() const
    {
        const uint32_t m = collapse80(data[0]) | collapse80(data[1]) << 4 | collapse80(data[2]) << 8 | collapse80(data[3]) << 12;
        return m & ~((1u << lead) - 1u);
    }
    __device__ __forceinline__ uint32_t mark_mask() const
    {
        return collapse80(mark[0]) | collapse80(mark[1]) << 4 | collapse80(mark[2]) << 8 | collapse80(mark[3]) << 12;
    }
    /// Index of the first byte of the lane flagged in `bad`, or 16.
    __device__ __forceinline__ uint32_t first_bad() const
    {
        const uint32_t m = collapse80(bad[0]) | collapse80(bad[1]) << 4 | collapse80(bad[2]) << 8 | collapse80(bad[3]) << 12;
        return m ? static_cast<uint32_t>(__ffs(m) - 1) : 16u;
    }
    __device__ __forceinline__ uint32_t byte_after(int i) const
    {
        return i < 15 ? (word[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xFFu : next;
    }
};

__device__ __forceinline__ LaneBytes classify(const FrontParams& P, uint32_t gpos)
{
    JG_GLOBAL const uint8_t* src = as_global(P.bytes);
    LaneBytes L;
    {
        typedef uint32_t V4 __attribute__((ext_vector_type(4)));
        const V4 v = *reinterpret_cast<JG_GLOBAL const V4*>(src + gpos); // the buffer has a spare window behind bytes_len
        L.word[0] = v[0]; L.word[1] = v[1]; L.word[2] = v[2]; L.word[3] = v[3];
    }
    // neighbours' bytes from the neighbouring lanes; the ends of a wave read them
    uint32_t prev = __shfl_up(L.word[3] >> 24, 1);
    L.next        = __shfl_down(L.word[0] & 0xFFu, 1);
    if (lane_id() == 0) prev = gpos > 0 ? src[gpos - 1] : 0u;
    if (lane_id() == 63) L.next = src[gpos + 16];
    L.lead = 0;
    if (gpos <= P.scan_begin || gpos + 17 > P.bytes_len) { // a lane at one of the two ends of the scan's bytes
        if (gpos <= P.scan_begin) prev = 0;
        if (gpos + 16 >= P.bytes_len) L.next = 0xFFu;
        L.lead = min(P.scan_begin > gpos ? P.scan_begin - gpos : 0u, 16u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t pos = gpos + 4 * j + k;
                if (pos < P.scan_begin) L.word[j] &= ~(0xFFu << (8 * k));
                if (pos >= P.bytes_len) L.word[j] |= 0xFFu << (8 * k);
            }
        }
    }
    uint32_t F[4], Z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        F[j] = bytes_ff(L.word[j]);
        Z[j] = bytes_zero(L.word[j]);
    }
    const uint32_t f_before = prev == 0xFFu ? 0x80000000u : 0u;
    const uint32_t f_after = L.next == 0xFFu ? 0x80u : 0u, z_after = L.next == 0u ? 0x80u : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t PF = of_previous_byte(F[j], j ? F[j - 1] : f_before);
        const uint32_t NF = of_next_byte(F[j], j < 3 ? F[j + 1] : f_after);
        const uint32_t NZ = of_next_byte(Z[j], j < 3 ? Z[j + 1] : z_after);
        L.data[j]         = (PF & Z[j]) | (~(PF | F[j]) & kHi80);
        L.mark[j]         = F[j] & ~(NF | NZ);
        L.bad[j]          = F[j] & PF & NZ;
    }
    return L;
}

/// Pass 1: data bytes and markers per window.
template <class FS>
__global__ __launch_bounds__(256) void front_count(FS src)
{
    const FrontParams P = src.get();
    if constexpr (FS::kCarriesJob) {
        if (blockIdx.x == 0) { // front_plan, three launches later, fills in the counts
            static_assert(sizeof(ScanJob) % 4 == 0, "copied as dwords");
            const uint32_t* from = reinterpret_cast<const uint32_t*>(&src.job);
            JG_GLOBAL uint32_t* to = reinterpret_cast<JG_GLOBAL uint32_t*>(as_global(P.job));
            for (uint32_t i = threadIdx.x; i < sizeof(ScanJob) / 4; i += blockDim.x) to[i] = from[i];
        }
    }
    if (blockIdx.x >= P.num_windows) return;
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_bad;
    if (threadIdx.x == 0) s_bad = 0xFFFFFFFFu;
    const uint32_t w    = blockIdx.x;
    const uint32_t gpos = w * kDestuffWin + threadIdx.x * 16;
    const LaneBytes L   = classify(P, gpos);
    uint32_t total_data = 0, total_mark = 0;
    block_excl_scan<4>(L.data_count(), s_wave, total_data); // (barriers inside: s_bad is initialised for everyone)
    block_excl_scan<4>(L.mark_count(), s_wave, total_mark);
    const uint32_t fb = L.first_bad();
    if (fb < 16u) atomicMin(&s_bad, gpos + fb);
    __syncthreads();
    if (threadIdx.x == 0) {
        as_global(P.win_data)[w]  = total_data;
        as_global(P.win_nmark)[w] = total_mark;
        as_global(P.win_bad)[w]   = s_bad;
    }
}

/// Pass 2, after the prefix sums over windows: every marker knows its ordinal in the scan; the first
/// expect_segments + 1 of them are recorded {position, data bytes of the scan in front of it}, and the first
/// one that is not RSTn -- the end of the scan -- is found with an atomic minimum.
template <class FS>
__global__ __launch_bounds__(256) void front_marks(FS src)
{
    const FrontParams P = src.get();
    if (blockIdx.x >= P.num_windows) return;
    __shared__ uint32_t s_wave[4];
    const uint32_t w    = blockIdx.x;
    const uint32_t gpos = w * kDestuffWin + threadIdx.x * 16;
    if (as_global(P.win_nmark)[w] == 0) return;
    const LaneBytes L = classify(P, gpos);
    uint32_t total_data = 0, total_mark = 0;
    const uint32_t data_before = as_global(P.win_prefix)[w] + block_excl_scan<4>(L.data_count(), s_wave, total_data);
    uint32_t ord               = as_global(P.mark_off)[w] + block_excl_scan<4>(L.mark_count(), s_wave, total_mark);
    uint32_t m                 = L.mark_mask();
    const uint32_t data_mask   = L.data_mask();
    while (m) {
        const int i = __ffs(m) - 1;
        m &= m - 1;
        if (ord <= P.expect_segments) {
            as_global(P.mk_pos)[ord] = gpos + i;
            as_global(P.mk_g)[ord]   = data_before + __popc(data_mask & ((1u << i) - 1u));
            const uint32_t code      = L.byte_after(i);
            if (code < 0xD0u || code > 0xD7u) atomicMin(P.status + 7, ord);
        }
        ++ord;
    }
}

constexpr int PL = 1024; // lanes of the planning workgroup

/// Exclusive prefix of in[0..n) into out[0..n], out[n] = total (in and out may be the same array).
__device__ uint32_t scan_array(JG_GLOBAL const uint32_t* in, JG_GLOBAL uint32_t* out, uint32_t n, uint32_t* s_wave)
{
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += PL) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan<PL / 64>(v, s_wave, total);
        if (i < n) out[i] = carry + ex;
        carry += total;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[n] = carry;
    __syncthreads();
    return carry;
}

/// Between the passes: prefix sums over the windows.
template <class FS>
__global__ __launch_bounds__(PL) void front_prefix(FS src)
{
    const FrontParams P = src.get();
    __shared__ uint32_t s_wave[PL / 64];
    scan_array(as_global(P.win_data), as_global(P.win_prefix), P.num_windows, s_wave);
    scan_array(as_global(P.win_nmark), as_global(P.mark_off), P.num_windows, s_wave);
    // first "FF FF 00" of the transferred bytes (the host walk refuses it inside the scan, jg_reader.cpp walk_scan)
    __shared__ uint32_t s_bad;
    if (threadIdx.x == 0) s_bad = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < P.num_windows; i += PL) bad = min(bad, as_global(P.win_bad)[i]);
    if (bad != 0xFFFFFFFFu) atomicMin(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        as_global(P.status)[6] = s_bad;
        as_global(P.status)[7] = 0xFFFFFFFFu; // ordinal of the terminating marker (front_marks)
    }
}

template <class FS>
__global__ __launch_bounds__(PL) void front_plan(FS src)
{
    const FrontParams P = src.get();
    __shared__ uint32_t s_wave[PL / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t Wn = P.num_windows, E = P.expect_segments, SB = P.subseq_bytes;
    JG_GLOBAL const uint32_t* WP  = as_global(P.win_prefix);
    JG_GLOBAL const uint32_t* mkp = as_global(P.mk_pos);
    JG_GLOBAL const uint32_t* mkg = as_global(P.mk_g);
    JG_GLOBAL uint32_t* cnt  = as_global(P.seg_cnt);
    JG_GLOBAL uint32_t* nch  = as_global(P.seg_nch);
    JG_GLOBAL Segment* segs  = as_global(P.segments);
    JG_GLOBAL uint32_t* stat = as_global(P.status);
    (void)Wn;

    const uint32_t T = stat[7]; // ordinal of the first marker that is not RSTn, among the first E + 1 markers
    uint32_t status  = 0;       // JPEGGPU_SUCCESS
    // JPEGGPU_INVALID_JPEG, as the host walk says in each of these cases (jg_reader.cpp walk_scan): no marker ends the
    // scan (or more restart markers than the geometry allows), segments that do not match the geometry, an
    // "FF FF 00" in front of the terminating marker
    if (T == 0xFFFFFFFFu) status = 2;
    else if (T + 1 != E) status = 2;
    else if (stat[6] < mkp[T]) status = 2;
    if (status != 0) {
        // nothing downstream may run on tables that were not built
        if (tid == 0) {
            JG_GLOBAL ScanJob* job = as_global(P.job);
            job->num_chunks        = 0;
            job->num_seq           = 0;
            job->num_tail_parts    = 0;
            job->num_mh_blocks     = 0;
            job->sp.num_subseq     = 0;
            job->sp.num_segments   = 0;
            job->ip.num_du         = 0;
            stat[0] = status;
            stat[1] = 0;
            stat[2] = 0;
            stat[3] = 0;
            stat[4] = 0;
        }
        return;
    }

    // segments: data bytes -> subsequences
    bool too_big = false, empty = false;
    for (uint32_t k = tid; k < E; k += PL) {
        const uint32_t g0 = k == 0 ? 0u : mkg[k - 1];
        const uint32_t d  = mkg[k] - g0;
        too_big |= d > (1u << 27); // bit positions are 32-bit
        empty |= d == 0;           // a restart interval holds at least one MCU: no bytes at all is not a JPEG (as the host walk)
        cnt[k] = (d + SB - 1) / SB;
        // chunks: the 4 KiB windows the segment's bytes touch (at least one, which carries the padding)
        const uint32_t b0 = k == 0 ? P.scan_begin : mkp[k - 1] + 2;
        const uint32_t b1 = mkp[k];
        const uint32_t last = b1 > b0 ? b1 - 1 : b0;
        nch[k] = last / kDestuffWin - b0 / kDestuffWin + 1;
    }
    const bool any_empty = __syncthreads_or(empty);
    if (__syncthreads_or(too_big) || any_empty) {
        if (tid == 0) {
            JG_GLOBAL ScanJob* job = as_global(P.job);
            job->num_chunks = job->num_seq = job->num_tail_parts = job->num_mh_blocks = 0;
            job->sp.num_subseq = job->sp.num_segments = 0;
            job->ip.num_du = 0;
            stat[0] = any_empty ? 2 : 4; // JPEGGPU_INVALID_JPEG / JPEGGPU_NOT_SUPPORTED
            stat[1] = stat[2] = stat[3] = stat[4] = 0;
        }
        return;
    }
    // subsequence offsets in place (cnt becomes the exclusive prefix; counts are recovered as differences)
    const uint32_t S = scan_array(cnt, cnt, E, s_wave);
    const uint32_t C = scan_array(nch, nch, E, s_wave);
    for (uint32_t k = tid; k < E; k += PL) {
        Segment sg;
        sg.subseq_offset = static_cast<int>(cnt[k]);
        sg.subseq_count  = static_cast<int>(cnt[k + 1] - cnt[k]);
        typedef uint32_t V2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<JG_GLOBAL V2*>(segs + k) = __builtin_bit_cast(V2, sg);
    }
    // destuff work list: chunk c belongs to the segment whose chunk range holds it
    for (uint32_t c = tid; c < C && c < P.max_chunks; c += PL) {
        uint32_t lo = 0, hi = E; // last k with nch[k] <= c
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (nch[mid] <= c) lo = mid;
            else hi = mid;
        }
        const uint32_t k  = lo;
        const uint32_t b0 = k == 0 ? P.scan_begin : mkp[k - 1] + 2;
        const uint32_t b1 = mkp[k];
        const uint32_t g0 = k == 0 ? 0u : mkg[k - 1];
        const uint32_t w  = b0 / kDestuffWin + (c - nch[k]);
        const uint32_t wb = w * kDestuffWin;
        DestuffChunk ck;
        ck.win_off  = wb;
        ck.begin    = b0 > wb ? b0 : wb;
        ck.end      = b1 < wb + kDestuffWin ? b1 : wb + kDestuffWin;
        if (ck.end < ck.begin) ck.end = ck.begin;
        ck.dst_off  = cnt[k] * SB + (ck.begin == b0 ? 0u : WP[w] - g0);
        ck.pad_end  = c + 1 == nch[k + 1] ? cnt[k + 1] * SB : 0u;
        ck.seg      = static_cast<int32_t>(k);
        ck.first    = c == nch[k] ? 1u : 0u;
        ck.reserved = 0;
        typedef uint32_t V8 __attribute__((ext_vector_type(8)));
        *reinterpret_cast<JG_GLOBAL V8*>(as_global(P.chunks) + c) = __builtin_bit_cast(V8, ck);
    }
    // tail-kernel parts: cut in front of a segment whose first subsequence opens a new block of
    // kTailPartSubseq subsequences
    JG_GLOBAL int* parts = as_global(P.tail_parts);
    uint32_t nparts      = 0;
    for (uint32_t base = 0; base < E; base += PL) {
        const uint32_t k = base + tid;
        bool cut         = false;
        if (k < E) cut = k == 0 || cnt[k] / kTailPartSubseq != cnt[k - 1] / kTailPartSubseq;
        uint32_t total;
        const uint32_t r = block_excl_scan<PL / 64>(cut ? 1u : 0u, s_wave, total);
        if (cut && nparts + r < P.max_parts) parts[nparts + r] = static_cast<int>(cnt[k]);
        nparts += total;
    }
    if (nparts >= P.max_parts) nparts = P.max_parts - 1;
    __syncthreads();
    // blocks of the multi-hypothesis chain walk (a scan without restart markers: ONE segment of S subsequences, which the
    // host could only bound): block b = subsequences [b * kMhMaxSegSubseq, ...), the first one opens the segment
    uint32_t mh_nb = 0;
    if (P.mh_blocks != nullptr && E == 1) {
        mh_nb = (S + kMhMaxSegSubseq - 1) / kMhMaxSegSubseq;
        if (mh_nb > P.max_mh_blocks) mh_nb = 0; // (cannot happen while S <= max_subseq: the plan sized the list from that bound)
        for (uint32_t b = tid; b < mh_nb; b += PL) {
            MhBlock blk;
            blk.first   = static_cast<int>(b * kMhMaxSegSubseq);
            blk.count   = static_cast<int>(S - b * kMhMaxSegSubseq < static_cast<uint32_t>(kMhMaxSegSubseq) ? S - b * kMhMaxSegSubseq : kMhMaxSegSubseq);
            blk.seg_end = static_cast<int>(S);
            blk.opens   = b == 0 ? 1 : 0;
            typedef uint32_t V4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<JG_GLOBAL V4*>(as_global(P.mh_blocks) + b) = __builtin_bit_cast(V4, blk);
        }
    }
    if (tid == 0) {
        parts[nparts]          = static_cast<int>(S);
        JG_GLOBAL ScanJob* job = as_global(P.job);
        const bool fits        = S <= P.max_subseq && C <= P.max_chunks;
        if (P.mh_blocks != nullptr) {
            job->num_mh_blocks = fits ? static_cast<int>(mh_nb) : 0;
            if (!fits || mh_nb == 0) job->sp.mh = 0; // no list: the sequence kernel speculates plainly
        }
        job->num_chunks        = fits ? static_cast<int>(C) : 0;
        const uint32_t seq     = static_cast<uint32_t>(job->sp.seq_subseq); // per job: a lone decode's sequences or a batch's (jg_defs.h)
        job->num_seq           = fits ? static_cast<int>((S + seq - 1) / seq) : 0;
        job->num_tail_parts    = fits ? static_cast<int>(nparts) : 0;
        job->sp.num_subseq     = fits ? static_cast<int>(S) : 0;
        job->sp.num_segments   = static_cast<int>(E);
        if (!fits) job->ip.num_du = 0;
        stat[0] = fits ? 0u : 3u; // JPEGGPU_INTERNAL_ERROR: bounds computed from the header were wrong
        stat[1] = S;
        stat[2] = E;
        stat[3] = C;
        stat[4] = nparts;
    }
}

} // namespace

hipError_t launch_front(const FrontParams& P, const ScanJob& job, hipStream_t stream)
{
    if (P.num_windows == 0) return hipErrorInvalidValue;
    const FrontOne src{P};
    front_count<<<P.num_windows, 256, 0, stream>>>(FrontOneJob{P, job});
    front_prefix<<<1, PL, 0, stream>>>(src);
    front_marks<<<P.num_windows, 256, 0, stream>>>(src);
    front_plan<<<1, PL, 0, stream>>>(src);
    return hipGetLastError();
}

hipError_t launch_front_batch(const FrontParams* d_params, int count, uint32_t max_windows, hipStream_t stream)
{
    if (count <= 0 || max_windows == 0) return hipErrorInvalidValue;
    const FrontMany src{d_params};
    const dim3 wide(max_windows, static_cast<uint32_t>(count)), one(1, static_cast<uint32_t>(count));
    front_count<<<wide, 256, 0, stream>>>(src);
    front_prefix<<<one, PL, 0, stream>>>(src);
    front_marks<<<wide, 256, 0, stream>>>(src);
    front_plan<<<one, PL, 0, stream>>>(src);
    return hipGetLastError();
}

} // namespace jg
