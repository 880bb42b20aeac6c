// jg_decoder.cpp -- per-image orchestration and the exported C ABI (include/jpeggpu/jpeggpu.h).
//
// Counterpart of the reference's src/jpeggpu.cpp:39-160 (argument checks, status strings) and
// src/decoder.cpp:67-354 (parse -> size -> transfer -> decode). Differences by design:
//   * one `plan` carves d_tmp for get_buffer_size / transfer / decode alike (the reference replays
//     decode_impl<false>, decoder.cpp:327-334);
//   * transfer ships the entropy-coded byte range plus ONE pinned table blob (2 copies instead of
//     ~10, and no post-EOI trailer; decoder.cpp:175-208);
//   * coefficients stay in stream order; there are no per-component coefficient planes, no transpose
//     pass and no DC pass (decoder.cpp:240-314).
#include "jg_front.hpp"
#include "jg_kernels.hpp"
#include "jg_reader.hpp"
#include "jg_selftest_data.h"

#include <jpeggpu/jpeggpu.h>
#include <jpeggpu/jpeggpu_ext.h>

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <new>
#include <vector>

namespace jg {

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/// Host staging memory for the table blob: page-locked when a HIP device is present (so the copy
/// enqueued by transfer is asynchronous), pageable otherwise (header parsing needs no GPU).
struct StagingBuffer {
    uint8_t* ptr  = nullptr;
    size_t cap    = 0;
    bool pinned   = false;

    bool reserve(size_t n)
    {
        if (n <= cap) return true;
        release();
        const size_t want = align_up(n + n / 2, 4096);
        void* p           = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess && p) {
            pinned = true;
        } else {
            (void)hipGetLastError();
            p      = std::malloc(want);
            pinned = false;
        }
        if (!p) return false;
        ptr = static_cast<uint8_t*>(p);
        cap = want;
        return true;
    }
    void release()
    {
        if (ptr) {
            if (pinned) (void)hipHostFree(ptr);
            else std::free(ptr);
        }
        ptr = nullptr;
        cap = 0;
    }
};

struct ScanPlan {
    // offsets inside the blob (and, shifted by off_blob, inside d_tmp)
    size_t blob_tables = 0, blob_tables_sync = 0, blob_segments = 0, blob_chunks = 0, blob_parts = 0;
    // offsets inside d_tmp
    size_t destuffed = 0, seg_idx = 0, st_p = 0, st_n = 0, st_cz = 0, st_dc01 = 0, st_dc23 = 0;
    size_t tails_n = 0, tails_dc01 = 0, tails_dc23 = 0, pending = 0, flow_list = 0, bnd_p = 0, bnd_cz = 0, fuse_ctl = 0;
    size_t sym = 0, du_tab = 0;
    size_t mh_p = 0, mh_cz = 0, mh_link = 0, mh_pool = 0, mh_known = 0; // multi-hypothesis speculation (jg_defs.h), if mh > 1
    size_t blob_mh_blocks = 0, mh_blk_exit = 0, mh_blk_entry = 0;        // its block-wise chain walk, if mh_blocks is not empty
    std::vector<jg::MhBlock> mh_blocks;
    int mh_blocks_device = 0;           // device-scanned scan without restart markers: capacity of the list jg_front.hip builds
    size_t d_mh_blocks = 0;             //   ... and where it sits in d_tmp
    int mh = 0, max_seg_subseq = 0;
    int num_seq = 0;
    // device-side front end (jg_front.hip): tables built on the device, scratch, the job and the status word
    size_t d_segments = 0, d_chunks = 0, d_parts = 0;
    size_t d_win_data = 0, d_win_nmark = 0, d_win_bad = 0, d_win_prefix = 0, d_mark_off = 0;
    size_t d_mk_pos = 0, d_mk_g = 0, d_seg_cnt = 0, d_seg_nch = 0, d_job = 0, d_status = 0;
    uint32_t num_windows = 0;
};

struct Plan {
    size_t off_bytes = 0, bytes_len = 0;
    size_t off_blob = 0, blob_size = 0;
    size_t blob_qtables = 0;
    size_t total = 0;
    ScanPlan scan[kMaxScans];
};

} // namespace

struct Decoder {
    Reader reader;
    Logger logger;
    StagingBuffer blob;
    Plan plan;
    const uint8_t* data = nullptr;
    size_t data_size    = 0;
    // Subsequence size: chosen PER IMAGE at parse_header (jg_reader.hpp, choose_subseq_bytes) from the scan's size, its
    // restart density and the call type -- `batched`: the decoder's images share their launches with others
    // (jpeggpu_ext_set_batched; jpeggpu_ext_decode_batch accepts any mix of sizes) -- unless the caller fixed one
    // (jpeggpu_ext_set_subsequence_bytes, JPEGGPU_SUBSEQ_BYTES). `subseq_bytes` is the size of the last parsed image.
    int subseq_request  = 0;     // 0: choose per image; else 32 / 64 / 128 / 256
    // About how many images of this kind share one jpeggpu_ext_decode_batch call (jpeggpu_ext_set_batch_hint; 0: decoded on
    // its own, jpeggpu_ext_set_batched(1): kBatchHintFull). `batched`: the plan is a batch's (no multi-hypothesis tables).
    int batch_hint      = 0;
    bool batched        = false;
    int seq_subseq_used = 0;     // subsequences per sequence of the last decode call built from this parse (0: none yet)
    bool mh_enabled     = true;  // JPEGGPU_MULTI_HYPOTHESIS=0 at startup: plain speculation for lone decodes as well
    int subseq_bytes    = 64;
    bool parsed         = false;
    int shard_rank = 0, shard_world = 1; // jpeggpu_ext_set_segment_shard
    int device_scan     = 0;     // jpeggpu_ext_set_device_scan: 0 off, 1 on (status via jpeggpu_ext_get_device_status), 2 on and checked by decode

    std::vector<ScanJob> jobs; // scratch of the last decode

    // optional stage timing (jpeggpu_ext_set_profiling): events recorded between the launches. A ring
    // of event sets so that several decodes can be in flight before the times are read back.
    static constexpr int kEventSets = 64;
    struct EventSet {
        std::vector<hipEvent_t> events;
        std::vector<int> stage; // stage that ENDS at event i (event 0 has none)
        size_t used = 0;
    };
    bool profiling = false;
    std::vector<EventSet> sets;
    int cur_set    = -1;
    int sets_valid = 0;
    void next_event_set();
    bool mark(int stage, hipStream_t stream);

    void make_plan();
    bool fill_blob();
};

void Decoder::make_plan()
{
    const Stream& s = reader.s;
    Plan p;
    // table blob
    size_t b       = 0;
    p.blob_qtables = b;
    b += align_up(sizeof(s.qtable), 256);
    for (int i = 0; i < s.num_scans; ++i) {
        const Scan& sc    = s.scans[i];
        ScanPlan& sp      = p.scan[i];
        sp.blob_tables    = b;
        b += align_up(sc.table_pack.size(), 256);
        sp.blob_tables_sync = b;
        b += align_up(sc.table_pack_sync.size(), 256);
        sp.blob_segments  = b;
        b += align_up(sc.segments.size() * sizeof(Segment), 256);
        sp.blob_chunks    = b;
        b += align_up(sc.chunks.size() * sizeof(DestuffChunk), 256);
        sp.blob_parts     = b;
        b += align_up(sc.tail_parts.size() * sizeof(int), 256);
        // Multi-hypothesis speculation (jg_defs.h) for an image decoded on its own: several data units per MCU. Segments
        // the chain walk can hold in LDS are walked whole; longer ones (a scan without restart markers is one segment) in
        // blocks whose descriptors travel with the blob.
        sp.mh = 0;
        sp.mh_blocks.clear();
        // What the speculation buys depends on how long a decoder that is off by some data units stays undetected: as long
        // as the units it confuses share their code tables. `run` = the longest run of consecutive data units of the MCU
        // with the same tables: 4 for 4:2:0 (Y Y Y Y), 2 for 4:2:2 (Y Y | Cb Cr) and 4:4:4 (Cb Cr). Sync stage of one 12 MP
        // image, speculation on / off (us, round 4): 4:2:0 with restart markers 137 / 294, without (block-wise walk)
        // 186 / 278; 4:2:2 117 / 138 and 161 / 133; 4:4:4 95 / 62 and 133 / 75; BASELINE configs[4] (4 components, runs of
        // 2, no restart markers) 204 / 140. So: runs of three and more always; runs of two only in the cheaper whole-segment
        // form and only with four units or more per MCU.
        int run = 1;
        {
            int du_tabs[2 * kMaxDuPerMcu], m = 0;
            for (int rep = 0; rep < 2; ++rep)
                for (int a = 0; a < sc.num_comp; ++a)
                    for (int k = 0; k < sc.comp[a].h * sc.comp[a].v && m < 2 * kMaxDuPerMcu; ++k) du_tabs[m++] = sc.comp[a].dc_id * 4 + sc.comp[a].ac_id;
            for (int i = 1, cur = 1; i < m; ++i) {
                cur = du_tabs[i] == du_tabs[i - 1] ? cur + 1 : 1;
                run = std::max(run, std::min(cur, sc.du_per_mcu));
            }
        }
        if (!batched && sc.du_per_mcu >= 2 && sc.du_per_mcu <= kMhMaxHyp && mh_enabled && run >= 2) {
            int longest = sc.device_walk ? kMhMaxSegSubseq : 0; // the device finds the segments: it falls back where one is longer
            for (const Segment& g : sc.segments) longest = std::max(longest, g.subseq_count);
            // a device-scanned scan without restart markers is ONE segment whose length the host can only bound: the
            // device builds the block list from what it finds (jg_front.hip, front_plan), sized here from the bound
            const bool device_blocks = sc.device_walk && s.restart_interval == 0;
            if (device_blocks) longest = std::max(sc.num_subseq, kMhMaxSegSubseq + 1);
            if (longest <= kMhMaxSegSubseq && run < 3 && sc.du_per_mcu < 4) longest = -1; // (runs of two: four units and more)
            if (longest > kMhMaxSegSubseq && run < 3) longest = -1;                       // (block-wise: runs of three and more)
            sp.mh_blocks_device = 0;
            if (longest < 0) {
            } else if (longest <= kMhMaxSegSubseq) {
                sp.mh             = sc.du_per_mcu;
                sp.max_seg_subseq = longest;
            } else if (device_blocks) {
                const int nb = (sc.num_subseq + kMhMaxSegSubseq - 1) / kMhMaxSegSubseq;
                if (nb <= kMhMaxBlocks) {
                    sp.mh               = sc.du_per_mcu;
                    sp.max_seg_subseq   = kMhMaxSegSubseq;
                    sp.mh_blocks_device = nb;
                }
            } else if (!sc.device_walk) {
                for (const Segment& g : sc.segments) {
                    for (int r = 0; r < g.subseq_count; r += kMhMaxSegSubseq)
                        sp.mh_blocks.push_back(MhBlock{g.subseq_offset + r, std::min(kMhMaxSegSubseq, g.subseq_count - r),
                                                       g.subseq_offset + g.subseq_count, r == 0 ? 1 : 0});
                }
                if (sp.mh_blocks.size() <= static_cast<size_t>(kMhMaxBlocks)) {
                    sp.mh             = sc.du_per_mcu;
                    sp.max_seg_subseq = kMhMaxSegSubseq;
                    sp.blob_mh_blocks = b;
                    b += align_up(sp.mh_blocks.size() * sizeof(MhBlock), 256);
                } else {
                    sp.mh_blocks.clear();
                }
            }
        }
    }
    p.blob_size = b;

    // device carve: transferred region first, at fixed places (reference decoder.cpp:116-155)
    size_t o    = 0;
    p.off_bytes = o;
    p.bytes_len = s.xfer_end - s.xfer_begin;
    o += align_up(p.bytes_len, kDestuffWin) + kDestuffWin; // whole windows are loaded
    p.off_blob = o;
    o += align_up(p.blob_size, 256);
    for (int i = 0; i < s.num_scans; ++i) {
        const Scan& sc = s.scans[i];
        ScanPlan& sp   = p.scan[i];
        const size_t S = static_cast<size_t>(sc.num_subseq);
        sp.num_seq     = static_cast<int>((S + kSeqSubseq - 1) / kSeqSubseq);
        sp.destuffed   = o;
        o += align_up(tiled_buffer_bytes(static_cast<uint32_t>(S), subseq_bytes, 96) + 256, 256); // whole tiles of padded rows, 96 rows spare
        sp.seg_idx = o;
        o += align_up(S * 4, 256);
        sp.st_p = o;
        o += align_up(S * 4, 256);
        sp.st_n = o;
        o += align_up(S * 4, 256);
        sp.st_cz = o;
        o += align_up(S * 4, 256);
        sp.st_dc01 = o;
        o += align_up(S * 4, 256);
        sp.st_dc23 = o;
        o += align_up(S * 4, 256);
        sp.pending = o;
        o += align_up(S, 256);
        sp.flow_list = o;
        o += align_up(S * 4, 256);
        sp.tails_n = o;
        o += align_up(static_cast<size_t>(sp.num_seq) * 4, 256);
        sp.tails_dc01 = o;
        o += align_up(static_cast<size_t>(sp.num_seq) * 4, 256);
        sp.tails_dc23 = o;
        o += align_up(static_cast<size_t>(sp.num_seq) * 4, 256);
        sp.bnd_p = o;
        o += align_up(static_cast<size_t>(sp.num_seq) * 4, 256);
        sp.bnd_cz = o;
        o += align_up(static_cast<size_t>(sp.num_seq) * 4, 256);
        sp.fuse_ctl = o; // control words of huff_tail_write
        o += align_up(fuse_ctl_words(static_cast<size_t>(sp.num_seq)) * 4, 256);
        if (sp.mh > 1) { // multi-hypothesis speculation (decided with the blob, above)
            const size_t N = S * static_cast<size_t>(sp.mh);
            sp.mh_p        = o;
            o += align_up(N * 4, 256);
            sp.mh_cz = o;
            o += align_up(N * 4, 256);
            sp.mh_link = o;
            o += align_up(N * 4, 256);
            sp.mh_pool = o;
            o += align_up((1 + static_cast<size_t>(mh_pool_entries(static_cast<uint32_t>(S)))) * sizeof(uint2_t), 256);
            sp.mh_known = o;
            o += align_up(S, 256);
            const size_t nblocks = sp.mh_blocks_device ? static_cast<size_t>(sp.mh_blocks_device) : sp.mh_blocks.size();
            if (nblocks) {
                sp.mh_blk_exit = o;
                o += align_up(nblocks * 64 * sizeof(uint16_t), 256);
                sp.mh_blk_entry = o;
                o += align_up(nblocks * sizeof(uint16_t), 256);
            }
            if (sp.mh_blocks_device) {
                sp.d_mh_blocks = o;
                o += align_up(nblocks * sizeof(MhBlock), 256);
            }
        }
        if (sc.device_walk) {
            const size_t E   = static_cast<size_t>(sc.expect_segments);
            sp.num_windows   = static_cast<uint32_t>(align_up(p.bytes_len, kDestuffWin) / kDestuffWin) - sc.front_win0;
            const size_t Wn  = sp.num_windows;
            const auto carve = [&](size_t& at, size_t bytes) {
                at = o;
                o += align_up(bytes, 256);
            };
            carve(sp.d_segments, E * sizeof(Segment));
            carve(sp.d_chunks, static_cast<size_t>(sc.max_chunks) * sizeof(DestuffChunk));
            carve(sp.d_parts, static_cast<size_t>(sc.max_tail_parts) * sizeof(int));
            carve(sp.d_win_data, Wn * 4);
            carve(sp.d_win_nmark, Wn * 4);
            carve(sp.d_win_bad, Wn * 4);
            carve(sp.d_win_prefix, (Wn + 1) * 4);
            carve(sp.d_mark_off, (Wn + 1) * 4);
            carve(sp.d_mk_pos, (E + 1) * 4);
            carve(sp.d_mk_g, (E + 1) * 4);
            carve(sp.d_seg_cnt, (E + 1) * 4);
            carve(sp.d_seg_nch, (E + 1) * 4);
            carve(sp.d_job, sizeof(ScanJob));
            carve(sp.d_status, 32);
        }
    }
    for (int i = 0; i < s.num_scans; ++i) {
        p.scan[i].sym = o; // symbol stream: a fixed region per subsequence
        o += align_up(sym_buffer_entries(static_cast<uint32_t>(s.scans[i].num_subseq), sym_region_entries(subseq_bytes)) * 2 + 256, 256);
        p.scan[i].du_tab = o;
        o += align_up(static_cast<size_t>(s.scans[i].num_du) * sizeof(uint2_t), 256);
    }
    p.total = o;
    plan             = p;
}

void Decoder::next_event_set()
{
    if (!profiling) return;
    if (sets.empty()) sets.resize(kEventSets);
    cur_set = (cur_set + 1) % kEventSets;
    sets[cur_set].used = 0;
    if (sets_valid < kEventSets) ++sets_valid;
}

bool Decoder::mark(int stage, hipStream_t stream)
{
    if (!profiling || cur_set < 0) return true;
    EventSet& es = sets[cur_set];
    if (es.used == es.events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return false;
        es.events.push_back(e);
        es.stage.push_back(0);
    }
    es.stage[es.used] = stage;
    return hipEventRecord(es.events[es.used++], stream) == hipSuccess;
}

bool Decoder::fill_blob()
{
    const Stream& s = reader.s;
    if (!blob.reserve(plan.blob_size)) return false;
    std::memset(blob.ptr, 0, plan.blob_size);
    std::memcpy(blob.ptr + plan.blob_qtables, s.qtable, sizeof(s.qtable));
    for (int i = 0; i < s.num_scans; ++i) {
        const Scan& sc     = s.scans[i];
        const ScanPlan& sp = plan.scan[i];
        std::memcpy(blob.ptr + sp.blob_tables, sc.table_pack.data(), sc.table_pack.size());
        std::memcpy(blob.ptr + sp.blob_tables_sync, sc.table_pack_sync.data(), sc.table_pack_sync.size());
        if (!sc.segments.empty())
            std::memcpy(blob.ptr + sp.blob_segments, sc.segments.data(), sc.segments.size() * sizeof(Segment));
        if (!sc.chunks.empty())
            std::memcpy(blob.ptr + sp.blob_chunks, sc.chunks.data(), sc.chunks.size() * sizeof(DestuffChunk));
        if (!sc.tail_parts.empty())
            std::memcpy(blob.ptr + sp.blob_parts, sc.tail_parts.data(), sc.tail_parts.size() * sizeof(int));
        if (!sp.mh_blocks.empty())
            std::memcpy(blob.ptr + sp.blob_mh_blocks, sp.mh_blocks.data(), sp.mh_blocks.size() * sizeof(MhBlock));
    }
    return true;
}

} // namespace jg

struct jpeggpu_decoder {
    jg::Decoder d;
};

using jg::Decoder;

namespace {

#define JG_CHECK_HIP(call)                                                                          \
    do {                                                                                            \
        const hipError_t err_ = (call);                                                             \
        if (err_ != hipSuccess) {                                                                   \
            d.logger.log("HIP error \"%s\" at: " __FILE__ ":%d\n", hipGetErrorString(err_), __LINE__); \
            return JPEGGPU_INTERNAL_ERROR;                                                          \
        }                                                                                           \
    } while (0)

jpeggpu_status do_transfer(Decoder& d, void* d_tmp, size_t tmp_size, hipStream_t stream)
{
    if (!d.parsed) return JPEGGPU_INVALID_ARGUMENT;
    if (!d_tmp || (reinterpret_cast<uintptr_t>(d_tmp) & 255)) return JPEGGPU_INVALID_ARGUMENT;
    if (tmp_size < d.plan.total) return JPEGGPU_INTERNAL_ERROR; // reference util.hpp:66-68
    uint8_t* base = static_cast<uint8_t*>(d_tmp);
    JG_CHECK_HIP(hipMemcpyAsync(
        base + d.plan.off_bytes, d.data + d.reader.s.xfer_begin, d.plan.bytes_len, hipMemcpyHostToDevice, stream));
    JG_CHECK_HIP(hipMemcpyAsync(
        base + d.plan.off_blob, d.blob.ptr, d.plan.blob_size, hipMemcpyHostToDevice, stream));
    return JPEGGPU_SUCCESS;
}

/// Validate the arguments of a decode and describe every scan of the image as a ScanJob.
/// `lone`: jpeggpu_decoder_decode (multi-hypothesis tables where the plan has them). `keep_flows`: every flow stays in its
/// sequence's workgroup (huff_sync_intra with re-packed flows; the tail kernel looks at sequence boundaries only) -- lone
/// decodes and batches too small to fill the chip; else the sequence kernel runs `max_intra_iters` iterations and marks
/// the rest for the tail kernel.
jpeggpu_status build_jobs(
    Decoder& d, const jpeggpu_img* img, void* d_tmp, size_t tmp_size, int max_intra_iters, bool lone, bool keep_flows, std::vector<jg::ScanJob>& jobs)
{
    using namespace jg;
    if (!d.parsed) return JPEGGPU_INVALID_ARGUMENT;
    const Stream& s = d.reader.s;
    for (int c = 0; c < s.num_comp; ++c) {
        if (!img->image[c] || img->pitch[c] < s.comp[c].size_x) return JPEGGPU_INVALID_ARGUMENT;
    }
    if (!d_tmp || (reinterpret_cast<uintptr_t>(d_tmp) & 255)) return JPEGGPU_INVALID_ARGUMENT;
    if (tmp_size < d.plan.total) return JPEGGPU_INTERNAL_ERROR;
    uint8_t* base    = static_cast<uint8_t*>(d_tmp);
    const Plan& plan = d.plan;
    uint8_t* blob    = base + plan.off_blob;

    for (int i = 0; i < s.num_scans; ++i) {
        const Scan& sc     = s.scans[i];
        const ScanPlan& pl = plan.scan[i];
        ScanJob job{};
        ScanParams& sp      = job.sp;
        sp.num_subseq       = sc.num_subseq;
        sp.num_segments     = static_cast<int>(sc.segments.size());
        sp.du_per_mcu       = sc.du_per_mcu;
        sp.num_comp         = sc.num_comp;
        sp.mcus_per_segment = sc.mcus_per_segment;
        sp.total_mcus       = sc.shard_mcus ? sc.shard_mcus : sc.mcus_x * sc.mcus_y; // of this decoder's share
        sp.subseq_words     = d.subseq_bytes / 4;
        sp.tab_bytes        = static_cast<uint32_t>(sc.table_pack.size());
        sp.max_intra_iters  = keep_flows ? kSeqLanes : max_intra_iters;
        sp.tail_marks       = keep_flows ? 0 : 1;
        sp.cursor_off       = sc.cursor_off;
        sp.tab_bytes_sync   = static_cast<uint32_t>(sc.table_pack_sync.size());
        sp.cursor_off_sync  = sc.cursor_off_sync;
        sp.mh               = lone ? pl.mh : 0; // the multi-hypothesis kernels run in front of a lone decode's sequence kernel only
        // a full batch's sequences are longer: one overlap lane (jg_defs.h); where every flow stays in the workgroup the 16
        // overlap lanes are what keeps the sequence boundaries from starting tail flows
        sp.seq_subseq       = keep_flows ? kSeqSubseq : kSeqSubseqBatch;
        d.seq_subseq_used   = sp.seq_subseq;
        job.mh_p            = reinterpret_cast<int*>(base + pl.mh_p);
        job.mh_cz           = reinterpret_cast<int*>(base + pl.mh_cz);
        job.mh_link         = reinterpret_cast<uint32_t*>(base + pl.mh_link);
        job.mh_pool         = reinterpret_cast<uint2_t*>(base + pl.mh_pool);
        job.mh_known        = base + pl.mh_known;
        job.num_mh_blocks   = lone ? static_cast<int>(pl.mh_blocks.size()) : 0;
        job.mh_blocks       = job.num_mh_blocks ? reinterpret_cast<const MhBlock*>(blob + pl.blob_mh_blocks) : nullptr;
        if (lone && pl.mh_blocks_device) { // the list the device builds (capacity here, the real count in its copy of the job)
            job.num_mh_blocks = pl.mh_blocks_device;
            job.mh_blocks     = reinterpret_cast<const MhBlock*>(base + pl.d_mh_blocks);
        }
        job.mh_blk_exit     = reinterpret_cast<uint16_t*>(base + pl.mh_blk_exit);
        job.mh_blk_entry    = reinterpret_cast<uint16_t*>(base + pl.mh_blk_entry);
        IdctParams& ip = job.ip;
        ip.num_du      = sc.num_du;
        ip.du_per_mcu  = sc.du_per_mcu;
        ip.mcus_x      = sc.mcus_x;
        ip.first_mcu   = sc.first_mcu;
        {
            const MagicDiv a = magic_div(static_cast<uint32_t>(sc.du_per_mcu)), b = magic_div(static_cast<uint32_t>(sc.mcus_x));
            ip.du_per_mcu_mul = a.mul, ip.du_per_mcu_shift = a.shift;
            ip.mcus_x_mul = b.mul, ip.mcus_x_shift = b.shift;
        }
        int du         = 0;
        for (int a = 0; a < sc.num_comp; ++a) {
            const ScanComponent& c = sc.comp[a];
            for (int y = 0; y < c.v; ++y) {
                for (int x = 0; x < c.h; ++x) { // row-major inside the MCU (T.81 A.2.3)
                    ip.du_comp[du] = static_cast<uint8_t>(a);
                    ip.du_dx[du]   = static_cast<uint8_t>(x);
                    ip.du_dy[du]   = static_cast<uint8_t>(y);
                    ++du;
                }
            }
            const Component& fc = s.comp[c.comp_idx];
            ip.comp_h[a]        = c.h;
            ip.comp_v[a]        = c.v;
            ip.size_x[a]        = fc.size_x;
            ip.size_y[a]        = fc.size_y;
            ip.pitch[a]         = img->pitch[c.comp_idx];
            ip.qidx[a]          = fc.qidx;
            ip.plane[a]         = img->image[c.comp_idx];
        }
        job.bytes      = base + plan.off_bytes;
        job.chunks     = reinterpret_cast<const DestuffChunk*>(blob + pl.blob_chunks);
        job.segments   = reinterpret_cast<const Segment*>(blob + pl.blob_segments);
        job.tables     = blob + pl.blob_tables;
        job.tables_sync = blob + pl.blob_tables_sync;
        job.qtables    = reinterpret_cast<const uint16_t*>(blob + plan.blob_qtables);
        job.destuffed  = base + pl.destuffed;
        job.seg_idx    = reinterpret_cast<int*>(base + pl.seg_idx);
        job.st_p       = reinterpret_cast<int*>(base + pl.st_p);
        job.st_n       = reinterpret_cast<int*>(base + pl.st_n);
        job.st_cz      = reinterpret_cast<int*>(base + pl.st_cz);
        job.st_dc01    = reinterpret_cast<uint32_t*>(base + pl.st_dc01);
        job.st_dc23    = reinterpret_cast<uint32_t*>(base + pl.st_dc23);
        job.pending    = base + pl.pending;
        job.bnd_p      = reinterpret_cast<int*>(base + pl.bnd_p);
        job.bnd_cz     = reinterpret_cast<int*>(base + pl.bnd_cz);
        job.flow_list  = reinterpret_cast<int*>(base + pl.flow_list);
        job.tail_parts = reinterpret_cast<const int*>(blob + pl.blob_parts);
        job.num_tail_parts = static_cast<int>(sc.tail_parts.size()) - 1;
        job.max_tail_part  = 0;
        job.fuse_ctl       = reinterpret_cast<uint32_t*>(base + pl.fuse_ctl);
        for (size_t k = 0; k + 1 < sc.tail_parts.size(); ++k)
            job.max_tail_part = std::max(job.max_tail_part, sc.tail_parts[k + 1] - sc.tail_parts[k]);
        job.tails_n    = reinterpret_cast<int*>(base + pl.tails_n);
        job.tails_dc01 = reinterpret_cast<uint32_t*>(base + pl.tails_dc01);
        job.tails_dc23 = reinterpret_cast<uint32_t*>(base + pl.tails_dc23);
        job.sym         = reinterpret_cast<uint16_t*>(base + pl.sym);
        job.du_tab      = reinterpret_cast<uint2_t*>(base + pl.du_tab);
        job.sym_region  = sym_region_entries(d.subseq_bytes);
        job.sym_entries = sym_buffer_entries(static_cast<uint32_t>(sc.num_subseq), job.sym_region);
        job.num_chunks = static_cast<int>(sc.chunks.size());
        job.num_seq    = static_cast<int>((static_cast<size_t>(sc.num_subseq) + sp.seq_subseq - 1) / sp.seq_subseq); // <= pl.num_seq, what the arrays are sized for
        if (sc.device_walk) {
            // tables built by jg_front.hip in device memory; the counts below are capacities (launch extents),
            // the device writes the real ones into its copy of the job
            job.chunks         = reinterpret_cast<const DestuffChunk*>(base + pl.d_chunks);
            job.segments       = reinterpret_cast<const Segment*>(base + pl.d_segments);
            job.tail_parts     = reinterpret_cast<const int*>(base + pl.d_parts);
            job.num_chunks     = sc.max_chunks;
            job.num_tail_parts = sc.max_tail_parts - 1;
            job.max_tail_part  = s.restart_interval ? kTailPartSubseq : (1 << 30); // lanes of the tail kernel
            sp.num_segments    = sc.expect_segments;
            // the device's tables count from the window that holds this scan's first byte (earlier scans lie in front)
            job.bytes          = base + plan.off_bytes + static_cast<size_t>(sc.front_win0) * kDestuffWin;
        }
        jobs.push_back(job);
    }
    return JPEGGPU_SUCCESS;
}

/// Index of the scan of a parsed image that the device walks (its last one), or -1.
int device_scan_index(const Decoder& d)
{
    const jg::Stream& s = d.reader.s;
    return s.num_scans > 0 && s.scans[s.num_scans - 1].device_walk ? s.num_scans - 1 : -1;
}

/// Parameters of the device-side front end for the device-walked scan `k` of a parsed image; `d_job` is the device copy
/// of its job.
jg::FrontParams front_params(const Decoder& d, void* d_tmp, jg::ScanJob* d_job, int k)
{
    using namespace jg;
    const Scan& sc     = d.reader.s.scans[k];
    const ScanPlan& pl = d.plan.scan[k];
    uint8_t* base      = static_cast<uint8_t*>(d_tmp);
    const size_t skip  = static_cast<size_t>(sc.front_win0) * kDestuffWin; // whole windows of earlier scans' bytes
    FrontParams P{};
    P.bytes           = base + d.plan.off_bytes + skip;
    P.bytes_len       = static_cast<uint32_t>(d.plan.bytes_len - skip);
    P.scan_begin      = static_cast<uint32_t>(sc.begin - d.reader.s.xfer_begin - skip);
    P.num_windows     = pl.num_windows;
    P.expect_segments = static_cast<uint32_t>(sc.expect_segments);
    P.subseq_bytes    = static_cast<uint32_t>(d.subseq_bytes);
    P.max_subseq      = static_cast<uint32_t>(sc.num_subseq);
    P.max_chunks      = static_cast<uint32_t>(sc.max_chunks);
    P.max_parts       = static_cast<uint32_t>(sc.max_tail_parts);
    const auto u32    = [&](size_t off) { return reinterpret_cast<uint32_t*>(base + off); };
    P.win_data   = u32(pl.d_win_data);
    P.win_nmark  = u32(pl.d_win_nmark);
    P.win_bad    = u32(pl.d_win_bad);
    P.win_prefix = u32(pl.d_win_prefix);
    P.mark_off   = u32(pl.d_mark_off);
    P.mk_pos     = u32(pl.d_mk_pos);
    P.mk_g       = u32(pl.d_mk_g);
    P.seg_cnt    = u32(pl.d_seg_cnt);
    P.seg_nch    = u32(pl.d_seg_nch);
    P.segments   = reinterpret_cast<Segment*>(base + pl.d_segments);
    P.chunks     = reinterpret_cast<DestuffChunk*>(base + pl.d_chunks);
    P.tail_parts = reinterpret_cast<int*>(base + pl.d_parts);
    P.job        = d_job;
    P.status     = u32(pl.d_status);
    P.mh_blocks     = pl.mh_blocks_device ? reinterpret_cast<MhBlock*>(base + pl.d_mh_blocks) : nullptr;
    P.max_mh_blocks = static_cast<uint32_t>(pl.mh_blocks_device);
    return P;
}

/// The status word of the device-side front end (synchronises `stream`); SUCCESS for host-walked images.
jpeggpu_status read_device_status(Decoder& d, const void* d_tmp, hipStream_t stream, jpeggpu_status* status)
{
    *status = JPEGGPU_SUCCESS;
    const int k = device_scan_index(d);
    if (k < 0) return JPEGGPU_SUCCESS; // the host walk has already judged the stream
    uint32_t word = 0;
    const uint8_t* src = static_cast<const uint8_t*>(d_tmp) + d.plan.scan[k].d_status;
    if (hipMemcpyAsync(&word, src, sizeof(word), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return JPEGGPU_INTERNAL_ERROR;
    *status = word <= JPEGGPU_INCOMPLETE_BITSTREAM ? static_cast<jpeggpu_status>(word) : JPEGGPU_INTERNAL_ERROR;
    return JPEGGPU_SUCCESS;
}

/// `may_block`: the checked mode of the device scan may wait for the stream (jpeggpu_decoder_decode); an item of a batch is
/// never waited for (jpeggpu_ext.h).
jpeggpu_status do_decode(Decoder& d, jpeggpu_img* img, void* d_tmp, size_t tmp_size, hipStream_t stream, bool may_block = true)
{
    using namespace jg;
    d.jobs.clear();
    // one image: latency matters, keep every flow inside the sequence's workgroup
    const jpeggpu_status st = build_jobs(d, img, d_tmp, tmp_size, jg::kSeqLanes, true, true, d.jobs);
    if (st != JPEGGPU_SUCCESS) return st;
    d.next_event_set();
    d.mark(-1, stream);
    if (const int k = device_scan_index(d); k >= 0) {
        // Device-side front end for the image's last scan: its job lives in device memory, front_plan fills in its counts,
        // and the stages read it from there, with launch extents from the header's upper bounds. The scans in front of
        // it (a file of several scans) were walked on the host: their jobs travel by value, one launch per stage for all
        // of them beside the device-walked scan's.
        ScanJob* d_job = reinterpret_cast<ScanJob*>(static_cast<uint8_t*>(d_tmp) + d.plan.scan[k].d_job);
        const FrontParams P = front_params(d, d_tmp, d_job, k);
        // the job travels as a kernel argument of the first front-end kernel, which stores it to d_job
        JG_CHECK_HIP(launch_front(P, d.jobs[k], stream));
        JobExtent extent;
        extend(extent, d.jobs[k]);
        for (int stage = 0; stage < kNumStages; ++stage) {
            if (stage == kStageSyncIntra) {
                for (int i = 0; i < k; ++i)
                    if (d.jobs[i].sp.mh > 1) JG_CHECK_HIP(launch_mh(d.jobs[i], nullptr, d.plan.scan[i].max_seg_subseq, stream));
                if (d.jobs[k].sp.mh > 1) JG_CHECK_HIP(launch_mh(d.jobs[k], d_job, d.plan.scan[k].max_seg_subseq, stream));
            }
            if (k > 0) JG_CHECK_HIP(launch_stage_scans(static_cast<Stage>(stage), d.jobs.data(), k, stream));
            JG_CHECK_HIP(launch_stage_device_job(static_cast<Stage>(stage), d_job, extent, stream));
            d.mark(stage, stream);
        }
        if (d.device_scan == 2 && may_block) {
            // Checked mode (JPEGGPU_DEVICE_SCAN=1 / 2 / checked in the environment of a caller that knows only the drop-in API):
            // what the host walk would have reported from parse_header is known on the device only now. Wait for
            // it and return it, as such a caller cannot ask for it; nothing of the device-walked scan was written to the
            // planes if it is not success.
            jpeggpu_status dev = JPEGGPU_SUCCESS;
            const jpeggpu_status rc = read_device_status(d, d_tmp, stream, &dev);
            return rc != JPEGGPU_SUCCESS ? rc : dev;
        }
        return JPEGGPU_SUCCESS;
    }
    for (size_t i = 0; i < d.jobs.size(); ++i) {
        const ScanJob& job = d.jobs[i];
        d.logger.log(
            "scan %d: %d chunks, %d subsequences of %d bytes, %d sequences, %d segments\n",
            static_cast<int>(i), job.num_chunks, job.sp.num_subseq, d.subseq_bytes, job.num_seq, job.sp.num_segments);
    }
    // every stage once, for all scans of the image at a time (they are independent: grid.y = scan)
    for (int stage = 0; stage < kNumStages; ++stage) {
        if (stage == kStageSyncIntra) {
            // multi-hypothesis speculation in front of the sequence kernel, which then starts from its table
            for (size_t i = 0; i < d.jobs.size(); ++i)
                if (d.jobs[i].sp.mh > 1) JG_CHECK_HIP(launch_mh(d.jobs[i], nullptr, d.plan.scan[i].max_seg_subseq, stream));
        }
        JG_CHECK_HIP(launch_stage_scans(static_cast<Stage>(stage), d.jobs.data(), static_cast<int>(d.jobs.size()), stream));
        d.mark(stage, stream);
    }
    return JPEGGPU_SUCCESS;
}

} // namespace

extern "C" {

const char* jpeggpu_get_status_string(enum jpeggpu_status stat)
{
    switch (stat) { // same strings as the reference, src/jpeggpu.cpp:41-60
    case JPEGGPU_SUCCESS: return "success";
    case JPEGGPU_INVALID_ARGUMENT: return "invalid argument";
    case JPEGGPU_INVALID_JPEG: return "invalid jpeg";
    case JPEGGPU_INTERNAL_ERROR: return "internal jpeggpu error";
    case JPEGGPU_NOT_SUPPORTED: return "jpeg is not supported";
    case JPEGGPU_OUT_OF_HOST_MEMORY: return "out of host memory";
    case JPEGGPU_INCOMPLETE_BITSTREAM: return "incomplete bitstream";
    }
    return "unknown status";
}

enum jpeggpu_status jpeggpu_decoder_startup(jpeggpu_decoder_t* decoder)
{
    if (!decoder) return JPEGGPU_INVALID_ARGUMENT;
    *decoder = new (std::nothrow) jpeggpu_decoder();
    if (*decoder == nullptr) return JPEGGPU_OUT_OF_HOST_MEMORY;
    if (const char* e = std::getenv("JPEGGPU_SUBSEQ_BYTES")) {
        const int v = std::atoi(e);
        if (jg::subseq_bytes_supported(v)) (*decoder)->d.subseq_request = v;
    }
    // A caller of the drop-in API alone can opt into the device-side marker scan through the environment (jpeggpu_ext.h).
    if (const char* e = std::getenv("JPEGGPU_MULTI_HYPOTHESIS")) (*decoder)->d.mh_enabled = std::atoi(e) != 0;
    if (const char* e = std::getenv("JPEGGPU_DEVICE_SCAN")) {
        // 1, 2 and "checked": the CHECKED mode (2 of jpeggpu_ext_set_device_scan) -- a caller that knows only the drop-in
        // API cannot ask for the device's verdict, so decode waits for it and returns it; "async": mode 1, decode stays
        // asynchronous and a refused scan shows as untouched planes. (Round 3 made "1" the asynchronous mode; a caller who
        // had set it for round 2's checked mode lost the error reporting without notice: ADVICE r3.)
        const int v = std::atoi(e);
        const bool off = std::strcmp(e, "0") == 0 || std::strcmp(e, "off") == 0 || e[0] == 0;
        (*decoder)->d.device_scan = std::strcmp(e, "async") == 0 ? 1 : (v == 1 || v == 2 || std::strcmp(e, "checked") == 0) ? 2 : 0;
        // a value nobody recognises must not silently mean "off" (ADVICE r4); the mode in force is logged at parse_header
        if ((*decoder)->d.device_scan == 0 && !off)
            std::fprintf(stderr, "jpeggpu: JPEGGPU_DEVICE_SCAN=\"%s\" is not one of 0, off, 1, 2, checked, async: the device scan stays off\n", e);
    }
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_set_logging(jpeggpu_decoder_t decoder, int do_logging)
{
    if (!decoder) return JPEGGPU_INVALID_ARGUMENT;
    decoder->d.logger.enabled = do_logging != 0;
    return JPEGGPU_SUCCESS;
}

int is_css_444(struct jpeggpu_subsampling css, int num_components)
{
    if (num_components < 1 || num_components > JPEGGPU_MAX_COMP) return 0;
    for (int c = 0; c < num_components; ++c) {
        if (css.x[c] != 1 || css.y[c] != 1) return 0;
    }
    return 1;
}

enum jpeggpu_status jpeggpu_decoder_parse_header(
    jpeggpu_decoder_t decoder, struct jpeggpu_img_info* img_info, const uint8_t* data, size_t size)
{
    if (!decoder || !img_info || !data) return JPEGGPU_INVALID_ARGUMENT;
    Decoder& d = decoder->d;
    d.parsed   = false;
    jpeggpu_status st;
    try {
        const int ask = d.subseq_request > 0 ? d.subseq_request : -d.batch_hint; // 0 / -N: chosen per image for N images per call
        st = d.reader.parse(data, size, ask, d.logger, d.device_scan != 0, d.shard_rank, d.shard_world);
        d.subseq_bytes = d.reader.subseq_bytes();
    } catch (const std::bad_alloc&) {
        return JPEGGPU_OUT_OF_HOST_MEMORY;
    }
    if (st != JPEGGPU_SUCCESS) return st;
    const jg::Stream& s = d.reader.s;
    if (d.device_scan)
        d.logger.log("device-side marker scan: %s (jpeggpu_ext_set_device_scan / JPEGGPU_DEVICE_SCAN)\n",
                     d.device_scan == 2 ? "checked -- jpeggpu_decoder_decode waits for the stream and returns the device's status" : "asynchronous");
    std::memset(img_info, 0, sizeof(*img_info));
    img_info->num_components = s.num_comp;
    for (int c = 0; c < s.num_comp; ++c) {
        img_info->sizes_x[c]       = s.comp[c].size_x;
        img_info->sizes_y[c]       = s.comp[c].size_y;
        img_info->subsampling.x[c] = s.comp[c].hs;
        img_info->subsampling.y[c] = s.comp[c].vs;
    }
    d.data      = data;
    d.data_size = size;
    // The IDCT addresses the 16-bit symbol stream with 32-bit BYTE offsets and the data-unit table holds 32-bit
    // entry indices (jg_kernels.hip, entry_at / prefetch): a scan whose stream would not fit them (from about
    // 400 MB of entropy-coded data at 64-byte subsequences) is refused here instead of gathering from wrapped offsets.
    for (int i = 0; i < s.num_scans; ++i) {
        const uint64_t entries = jg::sym_buffer_entries(static_cast<uint32_t>(s.scans[i].num_subseq), jg::sym_region_entries(d.subseq_bytes));
        if (entries * 2u >= (1ull << 32)) {
            d.logger.log("scan %d: %d subsequences of %d bytes need a symbol stream of %llu bytes (32-bit offsets)\n", i,
                         s.scans[i].num_subseq, d.subseq_bytes, static_cast<unsigned long long>(entries * 2u));
            return JPEGGPU_NOT_SUPPORTED;
        }
    }
    d.make_plan();
    if (!d.fill_blob()) return JPEGGPU_OUT_OF_HOST_MEMORY;
    d.seq_subseq_used = 0;
    d.parsed = true;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_decoder_get_buffer_size(jpeggpu_decoder_t decoder, size_t* tmp_size)
{
    if (!decoder || !tmp_size) return JPEGGPU_INVALID_ARGUMENT;
    if (!decoder->d.parsed) return JPEGGPU_INVALID_ARGUMENT;
    *tmp_size = decoder->d.plan.total;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_decoder_transfer(
    jpeggpu_decoder_t decoder, void* d_tmp, size_t tmp_size, jpeggpu_stream_t stream)
{
    if (!decoder) return JPEGGPU_INVALID_ARGUMENT;
    return do_transfer(decoder->d, d_tmp, tmp_size, stream);
}

enum jpeggpu_status jpeggpu_decoder_decode(
    jpeggpu_decoder_t decoder, struct jpeggpu_img* img, void* d_tmp, size_t tmp_size, jpeggpu_stream_t stream)
{
    if (!decoder || !img) return JPEGGPU_INVALID_ARGUMENT;
    return do_decode(decoder->d, img, d_tmp, tmp_size, stream);
}

enum jpeggpu_status jpeggpu_decoder_cleanup(jpeggpu_decoder_t decoder)
{
    if (!decoder) return JPEGGPU_INVALID_ARGUMENT;
    decoder->d.blob.release();
    for (auto& es : decoder->d.sets)
        for (hipEvent_t e : es.events) (void)hipEventDestroy(e);
    delete decoder;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_set_subsequence_bytes(jpeggpu_decoder_t decoder, int subseq_bytes)
{
    if (!decoder || (subseq_bytes != 0 && !jg::subseq_bytes_supported(subseq_bytes))) return JPEGGPU_INVALID_ARGUMENT;
    decoder->d.subseq_request = subseq_bytes; // 0: back to the per-image choice
    decoder->d.parsed         = false;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_set_batched(jpeggpu_decoder_t decoder, int batched)
{
    if (!decoder) return JPEGGPU_INVALID_ARGUMENT;
    return jpeggpu_ext_set_batch_hint(decoder, batched != 0 ? jg::kBatchHintFull : 0);
}

enum jpeggpu_status jpeggpu_ext_set_batch_hint(jpeggpu_decoder_t decoder, int images_per_call)
{
    if (!decoder || images_per_call < 0) return JPEGGPU_INVALID_ARGUMENT;
    decoder->d.batch_hint = images_per_call;
    // up to kLonePlanImages images per call get the lone decode's plan (multi-hypothesis tables, 64-byte subsequences):
    // jpeggpu_ext_decode_batch then decodes them one by one, as jpeggpu_decoder_decode would (jg::lone_plan)
    decoder->d.batched = !jg::lone_plan(images_per_call);
    decoder->d.parsed  = false;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_set_segment_shard(jpeggpu_decoder_t decoder, int rank, int world)
{
    if (!decoder || world < 1 || rank < 0 || rank >= world) return JPEGGPU_INVALID_ARGUMENT;
    decoder->d.shard_rank  = rank;
    decoder->d.shard_world = world;
    decoder->d.parsed      = false;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_get_shard_rows(jpeggpu_decoder_t decoder, int component, int* first_row, int* num_rows)
{
    if (!decoder || !first_row || !num_rows) return JPEGGPU_INVALID_ARGUMENT;
    const Decoder& d = decoder->d;
    if (!d.parsed) return JPEGGPU_INVALID_ARGUMENT;
    const jg::Stream& s = d.reader.s;
    if (component < 0 || component >= s.num_comp) return JPEGGPU_INVALID_ARGUMENT;
    const jg::Component& fc = s.comp[component];
    *first_row = 0;
    *num_rows  = fc.size_y;
    const jg::Scan& sc = s.scans[0];
    if (sc.total_segments == 0) return JPEGGPU_SUCCESS; // no shard: every row
    int v = 1;
    for (int a = 0; a < sc.num_comp; ++a)
        if (sc.comp[a].comp_idx == component) v = sc.comp[a].v;
    const int row0 = sc.first_mcu / sc.mcus_x * 8 * v, row1 = (sc.first_mcu + sc.shard_mcus) / sc.mcus_x * 8 * v;
    *first_row = std::min(row0, fc.size_y);
    *num_rows  = std::min(row1, fc.size_y) - *first_row;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_set_profiling(jpeggpu_decoder_t decoder, int enable)
{
    if (!decoder) return JPEGGPU_INVALID_ARGUMENT;
    decoder->d.profiling  = enable != 0;
    decoder->d.cur_set    = -1;
    decoder->d.sets_valid = 0;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_get_stage_ms(jpeggpu_decoder_t decoder, float* ms)
{
    if (!decoder || !ms) return JPEGGPU_INVALID_ARGUMENT;
    Decoder& d = decoder->d;
    for (int i = 0; i < JPEGGPU_EXT_NUM_STAGES; ++i) ms[i] = 0.f;
    if (!d.profiling || d.sets_valid == 0) return JPEGGPU_INVALID_ARGUMENT;
    // mean over the decodes recorded since profiling was (re-)enabled, at most the last kEventSets
    for (int k = 0; k < d.sets_valid; ++k) {
        const Decoder::EventSet& es = d.sets[k];
        for (size_t i = 1; i < es.used; ++i) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, es.events[i - 1], es.events[i]) != hipSuccess) return JPEGGPU_INTERNAL_ERROR;
            const int st = es.stage[i];
            if (st >= 0 && st < JPEGGPU_EXT_NUM_STAGES) ms[st] += t;
        }
    }
    for (int i = 0; i < JPEGGPU_EXT_NUM_STAGES; ++i) ms[i] /= static_cast<float>(d.sets_valid);
    decoder->d.cur_set    = -1; // start a new measurement window
    decoder->d.sets_valid = 0;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_get_layout(jpeggpu_decoder_t decoder, struct jpeggpu_ext_layout* out)
{
    if (!decoder || !out) return JPEGGPU_INVALID_ARGUMENT;
    const Decoder& d = decoder->d;
    if (!d.parsed) return JPEGGPU_INVALID_ARGUMENT;
    const jg::Stream& s = d.reader.s;
    std::memset(out, 0, sizeof(*out));
    out->subsequence_bytes = d.subseq_bytes;
    // of the last decode call built from this parse, whichever call it was (jpeggpu_decoder_decode and small batches: 240, a
    // full batch: 255); before any: what a call of the kind the decoder was set up for would use
    out->subsequences_per_sequence = d.seq_subseq_used ? d.seq_subseq_used : d.batched ? jg::kSeqSubseqBatch : jg::kSeqSubseq;
    out->num_scans         = s.num_scans;
    out->transferred_bytes = d.plan.bytes_len;
    out->blob_bytes        = d.plan.blob_size;
    out->off_bytes         = d.plan.off_bytes;
    out->off_qtables       = d.plan.off_blob + d.plan.blob_qtables;
    out->shard_rank        = d.shard_rank;
    out->shard_world       = d.shard_world;
    for (int i = 0; i < s.num_scans; ++i) {
        const jg::Scan& sc         = s.scans[i];
        const jg::ScanPlan& pl     = d.plan.scan[i];
        jpeggpu_ext_scan_layout& o = out->scans[i];
        o.num_components           = sc.num_comp;
        for (int k = 0; k < sc.num_comp; ++k) o.component_idx[k] = sc.comp[k].comp_idx;
        o.off_state_dc01 = pl.st_dc01;
        o.off_state_dc23 = pl.st_dc23;
        o.num_subsequences   = sc.num_subseq;
        o.num_segments       = static_cast<int>(sc.segments.size());
        o.num_sequences      = static_cast<int>((static_cast<size_t>(sc.num_subseq) + out->subsequences_per_sequence - 1) / out->subsequences_per_sequence);
        o.num_data_units     = sc.num_du;
        o.data_units_per_mcu = sc.du_per_mcu;
        o.num_chunks         = static_cast<int>(sc.chunks.size());
        o.off_segments       = d.plan.off_blob + pl.blob_segments;
        o.off_chunks         = d.plan.off_blob + pl.blob_chunks;
        o.off_destuffed      = pl.destuffed;
        o.off_segment_index  = pl.seg_idx;
        o.off_state_p        = pl.st_p;
        o.off_state_n        = pl.st_n;
        o.off_state_cz       = pl.st_cz;
        o.off_symbols        = pl.sym;
        o.off_du_table       = pl.du_tab;
        o.symbol_region_entries = static_cast<int>(jg::sym_region_entries(d.subseq_bytes));
        o.device_scan           = sc.device_walk ? 1 : 0;
        o.hypotheses            = pl.mh;
        o.hypothesis_blocks     = pl.mh_blocks_device ? pl.mh_blocks_device : static_cast<int>(pl.mh_blocks.size());
        if (sc.device_walk) {
            o.num_segments      = sc.expect_segments;
            o.num_chunks        = sc.max_chunks;
            o.off_segments      = pl.d_segments;
            o.off_chunks        = pl.d_chunks;
            o.off_device_status = pl.d_status;
        }
    }
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_set_device_scan(jpeggpu_decoder_t decoder, int enable)
{
    if (!decoder || enable < 0 || enable > 2) return JPEGGPU_INVALID_ARGUMENT;
    decoder->d.device_scan = enable;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_get_device_status(
    jpeggpu_decoder_t decoder, const void* d_tmp, jpeggpu_stream_t stream, enum jpeggpu_status* status)
{
    if (!decoder || !d_tmp || !status) return JPEGGPU_INVALID_ARGUMENT;
    Decoder& d = decoder->d;
    if (!d.parsed) return JPEGGPU_INVALID_ARGUMENT;
    return read_device_status(d, d_tmp, stream, status);
}

struct jpeggpu_batch {
    static constexpr int kRing = 4;
    int max_jobs = 0;
    uint8_t* staging[kRing]     = {}; // ScanJob[n], then FrontParams[device-scanned images]
    hipEvent_t copied[kRing]    = {};
    bool in_use[kRing]          = {};
    int next                    = 0;
    // Flow iterations inside the sequence kernel of a batch. One: speculate + verify there, the rest in the tail kernel. More
    // were measured with the survivors re-packed into one wave per workgroup (round 4): 2 / 3 / 8 iterations take the tail
    // kernel from 364 to 204 / 70 / 14 us per 64 images and the sequence kernel from 680 to 937 / 1109 / 1138 -- a
    // workgroup keeps its 22 KB of tables in LDS while one of its four waves works, and LDS is what bounds the kernel.
    int sync_iters              = 1;
    bool sync_iters_set         = false; // jpeggpu_ext_batch_set_sync_iterations was called: the caller's cap, whatever the call's size
    // Calls of fewer subsequences than this keep every flow in the sequence kernel (decode_batch_impl).
    long long keep_flows_below  = jg::kKeepFlowsBelowSubseq;
    // Full batches: the tail kernel's parts and the write pass's sequences as one launch (jg_kernels.hip: huff_tail_write).
    bool fuse_tail_write        = true;
    // A caller with ONE stream leaves the GPU idle while the latency-bound tail kernel runs (a fifth of a
    // batch's time). With overlap > 1 the jobs are split into that many parts, part 0 on the caller's
    // stream and the others on internal streams forked from and joined back into it with events.
    static constexpr int kMaxOverlap = 4;
    int overlap                 = 1;
    hipStream_t aux[kMaxOverlap - 1] = {};
    hipEvent_t joined[kMaxOverlap - 1] = {};
    std::vector<jg::ScanJob> jobs;
    std::vector<jg::FrontParams> fronts;
    std::vector<int> order, group_begin; // scratch of decode_batch: items by subsequence size, job ranges of the sizes
    struct Part {                        // ... and the parts of the job array, one launch per stage each
        int begin, end, way;
        jg::JobExtent extent;
    };
    std::vector<Part> parts;
    // optional stage timing, same contract as the decoder's
    bool profiling = false;
    std::vector<std::vector<hipEvent_t>> sets; // ring of kNumStages + 1 events
    int cur_set = -1, sets_valid = 0;
};

size_t jpeggpu_ext_batch_scratch_size(int max_scans)
{
    return static_cast<size_t>(max_scans > 0 ? max_scans : 0) * (sizeof(jg::ScanJob) + sizeof(jg::FrontParams));
}

enum jpeggpu_status jpeggpu_ext_batch_create(jpeggpu_batch_t* batch, int max_scans)
{
    if (!batch || max_scans <= 0) return JPEGGPU_INVALID_ARGUMENT;
    jpeggpu_batch* b = new (std::nothrow) jpeggpu_batch();
    if (!b) return JPEGGPU_OUT_OF_HOST_MEMORY;
    b->max_jobs = max_scans;
    if (const char* e = std::getenv("JPEGGPU_EXP_KEEP_FLOWS_BELOW")) b->keep_flows_below = std::atoll(e); // experiments (tools/probe/batch_curve.py)
    if (const char* e = std::getenv("JPEGGPU_FUSE_TAIL_WRITE")) b->fuse_tail_write = std::atoi(e) != 0;
    for (int r = 0; r < jpeggpu_batch::kRing; ++r) {
        void* p = nullptr;
        if (hipHostMalloc(&p, jpeggpu_ext_batch_scratch_size(max_scans), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&b->copied[r], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            jpeggpu_ext_batch_destroy(b);
            return JPEGGPU_INTERNAL_ERROR; // the batch path needs a device: no fallback
        }
        b->staging[r] = static_cast<uint8_t*>(p);
    }
    *batch = b;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_batch_destroy(jpeggpu_batch_t batch)
{
    if (!batch) return JPEGGPU_INVALID_ARGUMENT;
    for (int w = 0; w < jpeggpu_batch::kMaxOverlap - 1; ++w) {
        if (batch->aux[w]) {
            (void)hipStreamSynchronize(batch->aux[w]);
            (void)hipStreamDestroy(batch->aux[w]);
        }
        if (batch->joined[w]) (void)hipEventDestroy(batch->joined[w]);
    }
    for (int r = 0; r < jpeggpu_batch::kRing; ++r) {
        if (batch->staging[r]) (void)hipHostFree(batch->staging[r]);
        if (batch->copied[r]) (void)hipEventDestroy(batch->copied[r]);
    }
    for (auto& set : batch->sets)
        for (hipEvent_t e : set) (void)hipEventDestroy(e);
    delete batch;
    return JPEGGPU_SUCCESS;
}

static enum jpeggpu_status decode_batch_impl(
    jpeggpu_batch_t batch,
    const struct jpeggpu_ext_batch_item* items,
    int num_items,
    void* d_scratch,
    size_t scratch_size,
    jpeggpu_stream_t stream);

enum jpeggpu_status jpeggpu_ext_decode_batch(
    jpeggpu_batch_t batch,
    const struct jpeggpu_ext_batch_item* items,
    int num_items,
    void* d_scratch,
    size_t scratch_size,
    jpeggpu_stream_t stream)
{
    try { // the scratch vectors of the batch grow with its first calls: no exception crosses the C ABI
        return decode_batch_impl(batch, items, num_items, d_scratch, scratch_size, stream);
    } catch (const std::bad_alloc&) {
        return JPEGGPU_OUT_OF_HOST_MEMORY;
    }
}

static enum jpeggpu_status decode_batch_impl(
    jpeggpu_batch_t batch,
    const struct jpeggpu_ext_batch_item* items,
    int num_items,
    void* d_scratch,
    size_t scratch_size,
    jpeggpu_stream_t stream)
{
    if (!batch || !items || num_items < 0 || !d_scratch) return JPEGGPU_INVALID_ARGUMENT;
    if (num_items == 0) return JPEGGPU_SUCCESS;
    batch->jobs.clear();
    batch->fronts.clear();
    jg::ScanJob* d_jobs_rw = static_cast<jg::ScanJob*>(d_scratch);
    uint32_t front_windows = 0;
    // One kernel variant per launch: the items are taken in the order of their subsequence size (chosen per image at
    // parse_header unless the caller fixed it), and every size is a group of launches of its own.
    for (int i = 0; i < num_items; ++i)
        if (!items[i].decoder || !items[i].img) return JPEGGPU_INVALID_ARGUMENT;
    // A call of one or two images planned as lone decodes (jpeggpu_ext_set_batch_hint): decoded one by one with the lone
    // decode's kernels, multi-hypothesis speculation included -- the chip is empty either way.
    {
        // (with the batch's stage timing on, the call takes the batch's kernels: its events sit between THOSE launches)
        bool all_lone = num_items <= jg::kLonePlanImages && !batch->profiling;
        for (int i = 0; i < num_items && all_lone; ++i) all_lone = !items[i].decoder->d.batched;
        if (all_lone) {
            for (int i = 0; i < num_items; ++i) {
                const jpeggpu_status st = do_decode(items[i].decoder->d, items[i].img, items[i].d_tmp, items[i].tmp_size, stream, false);
                if (st != JPEGGPU_SUCCESS) return st;
            }
            return JPEGGPU_SUCCESS;
        }
    }
    // Does the call fill the chip? A launch of fewer than kKeepFlowsBelowSubseq subsequences does not: its sequence kernel
    // keeps every flow in the workgroup (the lone decode's kernel, one job per blockIdx.y) and the tail kernel has only
    // the sequence boundaries to look at; a full batch runs one flow iteration there and leaves the rest to the tail kernel,
    // whose latency other launches hide.
    bool keep_flows = false;
    {
        long long total_subseq = 0;
        for (int i = 0; i < num_items; ++i) {
            const jg::Stream& s = items[i].decoder->d.reader.s;
            if (!items[i].decoder->d.parsed) return JPEGGPU_INVALID_ARGUMENT;
            for (int k = 0; k < s.num_scans; ++k) total_subseq += s.scans[k].num_subseq;
        }
        keep_flows = batch->sync_iters_set ? false : total_subseq < batch->keep_flows_below;
    }
    std::vector<int>& order = batch->order;
    order.resize(static_cast<size_t>(num_items));
    for (int i = 0; i < num_items; ++i) order[static_cast<size_t>(i)] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[a].decoder->d.subseq_bytes > items[b].decoder->d.subseq_bytes; });
    std::vector<int>& group_begin = batch->group_begin; // job index at which each size group starts, plus the end
    group_begin.clear();
    int subseq_bytes = 0;
    for (int k = 0; k < num_items; ++k) {
        const jpeggpu_ext_batch_item& it = items[order[static_cast<size_t>(k)]];
        if (it.decoder->d.subseq_bytes != subseq_bytes) {
            subseq_bytes = it.decoder->d.subseq_bytes;
            group_begin.push_back(static_cast<int>(batch->jobs.size()));
        }
        const size_t first_job = batch->jobs.size();
        const jpeggpu_status st = build_jobs(it.decoder->d, it.img, it.d_tmp, it.tmp_size, batch->sync_iters, false, keep_flows, batch->jobs);
        if (st != JPEGGPU_SUCCESS) return st;
        if (const int dk = device_scan_index(it.decoder->d); dk >= 0) {
            // device-side front end (jpeggpu_ext_set_device_scan): the counts of this job are filled in on the device
            batch->fronts.push_back(front_params(it.decoder->d, it.d_tmp, d_jobs_rw + first_job + static_cast<size_t>(dk), dk));
            front_windows = std::max(front_windows, batch->fronts.back().num_windows);
        }
    }
    group_begin.push_back(static_cast<int>(batch->jobs.size()));
    const int n         = static_cast<int>(batch->jobs.size());
    const int nf        = static_cast<int>(batch->fronts.size());
    const size_t jbytes = sizeof(jg::ScanJob) * static_cast<size_t>(n), fbytes = sizeof(jg::FrontParams) * static_cast<size_t>(nf);
    if (n > batch->max_jobs || scratch_size < jbytes + fbytes) return JPEGGPU_INVALID_ARGUMENT;
    const int r = batch->next;
    batch->next = (r + 1) % jpeggpu_batch::kRing;
    // the staging buffer may still be the source of a copy enqueued kRing batches ago
    if (batch->in_use[r] && hipEventSynchronize(batch->copied[r]) != hipSuccess) return JPEGGPU_INTERNAL_ERROR;
    std::memcpy(batch->staging[r], batch->jobs.data(), jbytes);
    if (nf) std::memcpy(batch->staging[r] + jbytes, batch->fronts.data(), fbytes);
    const jg::ScanJob* d_jobs = static_cast<const jg::ScanJob*>(d_scratch);
    std::vector<hipEvent_t>* ev = nullptr;
    if (batch->profiling) {
        if (batch->sets.empty()) batch->sets.resize(64);
        batch->cur_set = (batch->cur_set + 1) % 64;
        if (batch->sets_valid < 64) ++batch->sets_valid;
        ev = &batch->sets[batch->cur_set];
        while (ev->size() < static_cast<size_t>(jg::kNumStages) + 1) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return JPEGGPU_INTERNAL_ERROR;
            ev->push_back(e);
        }
        (void)hipEventRecord((*ev)[0], stream);
    }
    if (hipMemcpyAsync(d_scratch, batch->staging[r], jbytes + fbytes, hipMemcpyHostToDevice, stream) != hipSuccess) return JPEGGPU_INTERNAL_ERROR;
    if (nf && jg::launch_front_batch(reinterpret_cast<const jg::FrontParams*>(static_cast<const uint8_t*>(d_scratch) + jbytes), nf,
                                     front_windows, stream) != hipSuccess) {
        (void)hipGetLastError();
        return JPEGGPU_INTERNAL_ERROR;
    }
    // the staging buffer is free again, and the parts below may start: the job array is complete
    if (hipEventRecord(batch->copied[r], stream) != hipSuccess) return JPEGGPU_INTERNAL_ERROR;
    batch->in_use[r] = true;
    // parts of the job array: every size group is cut into up to `overlap` contiguous parts of at least 4 jobs each,
    // part w of every group on stream w
    const int ways = batch->overlap;
    hipStream_t part_stream[jpeggpu_batch::kMaxOverlap];
    part_stream[0] = stream;
    for (int w = 1; w < ways; ++w) {
        if (!batch->aux[w - 1]) {
            if (hipStreamCreateWithFlags(&batch->aux[w - 1], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&batch->joined[w - 1], hipEventDisableTiming) != hipSuccess)
                return JPEGGPU_INTERNAL_ERROR;
        }
        part_stream[w] = batch->aux[w - 1];
        if (hipStreamWaitEvent(part_stream[w], batch->copied[r], 0) != hipSuccess) return JPEGGPU_INTERNAL_ERROR; // fork
    }
    typedef jpeggpu_batch::Part Part;
    std::vector<Part>& parts = batch->parts;
    parts.clear();
    for (size_t g = 0; g + 1 < group_begin.size(); ++g) {
        const int a = group_begin[g], b = group_begin[g + 1];
        int gw = ways;
        while (gw > 1 && (b - a) / gw < 4) --gw;
        for (int w = 0; w < gw; ++w) {
            Part p{a + static_cast<int>(static_cast<long long>(b - a) * w / gw), a + static_cast<int>(static_cast<long long>(b - a) * (w + 1) / gw), w, jg::JobExtent{}};
            for (int j = p.begin; j < p.end; ++j) jg::extend(p.extent, batch->jobs[static_cast<size_t>(j)]);
            p.extent.repack_flows = keep_flows;
            p.extent.fuse_tail_write = batch->fuse_tail_write && !batch->sync_iters_set;
            if (p.end > p.begin) parts.push_back(p);
        }
    }
    for (int stage = 0; stage < jg::kNumStages; ++stage) {
        for (const Part& p : parts) {
            if (jg::launch_stage_batch(static_cast<jg::Stage>(stage), d_jobs + p.begin, p.end - p.begin, p.extent, part_stream[p.way]) != hipSuccess) {
                (void)hipGetLastError();
                return JPEGGPU_INTERNAL_ERROR;
            }
        }
        if (ev) (void)hipEventRecord((*ev)[stage + 1], stream); // stage times are those of the caller's stream
    }
    for (int w = 1; w < ways; ++w) { // join: the caller's stream completes when every part has
        if (hipEventRecord(batch->joined[w - 1], part_stream[w]) != hipSuccess ||
            hipStreamWaitEvent(stream, batch->joined[w - 1], 0) != hipSuccess)
            return JPEGGPU_INTERNAL_ERROR;
    }
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_batch_set_overlap(jpeggpu_batch_t batch, int parts)
{
    if (!batch || parts < 1 || parts > jpeggpu_batch::kMaxOverlap) return JPEGGPU_INVALID_ARGUMENT;
    batch->overlap = parts;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_batch_set_fused_tail(jpeggpu_batch_t batch, int enable)
{
    if (!batch) return JPEGGPU_INVALID_ARGUMENT;
    batch->fuse_tail_write = enable != 0;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_fused_tail_timeouts(unsigned int* count)
{
    if (!count) return JPEGGPU_INVALID_ARGUMENT;
    return jg::read_fuse_timeouts(count) == hipSuccess ? JPEGGPU_SUCCESS : JPEGGPU_INTERNAL_ERROR;
}

enum jpeggpu_status jpeggpu_ext_batch_set_sync_iterations(jpeggpu_batch_t batch, int iterations)
{
    if (!batch || iterations < 1) return JPEGGPU_INVALID_ARGUMENT; // the first flow iteration supplies n and the DC sums
    batch->sync_iters     = iterations;
    batch->sync_iters_set = true;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_batch_set_profiling(jpeggpu_batch_t batch, int enable)
{
    if (!batch) return JPEGGPU_INVALID_ARGUMENT;
    batch->profiling  = enable != 0;
    batch->cur_set    = -1;
    batch->sets_valid = 0;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_batch_get_stage_ms(jpeggpu_batch_t batch, float* ms)
{
    if (!batch || !ms) return JPEGGPU_INVALID_ARGUMENT;
    for (int i = 0; i < JPEGGPU_EXT_NUM_STAGES; ++i) ms[i] = 0.f;
    if (!batch->profiling || batch->sets_valid == 0) return JPEGGPU_INVALID_ARGUMENT;
    for (int k = 0; k < batch->sets_valid; ++k) {
        for (int st = 0; st < jg::kNumStages; ++st) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, batch->sets[k][st], batch->sets[k][st + 1]) != hipSuccess) return JPEGGPU_INTERNAL_ERROR;
            ms[st] += t;
        }
    }
    for (int i = 0; i < JPEGGPU_EXT_NUM_STAGES; ++i) ms[i] /= static_cast<float>(batch->sets_valid);
    batch->cur_set    = -1;
    batch->sets_valid = 0;
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_upsample_planes(
    const struct jpeggpu_img_info* info,
    const struct jpeggpu_img* src,
    struct jpeggpu_img* dst,
    int width,
    int height,
    jpeggpu_stream_t stream)
{
    if (!info || !src || !dst || width <= 0 || height <= 0) return JPEGGPU_INVALID_ARGUMENT;
    const int nc = info->num_components;
    if (nc < 1 || nc > JPEGGPU_MAX_COMP) return JPEGGPU_INVALID_ARGUMENT;
    int sx_max = 0, sy_max = 0;
    for (int c = 0; c < nc; ++c) {
        if (info->subsampling.x[c] < 1 || info->subsampling.y[c] < 1) return JPEGGPU_INVALID_ARGUMENT;
        sx_max = info->subsampling.x[c] > sx_max ? info->subsampling.x[c] : sx_max;
        sy_max = info->subsampling.y[c] > sy_max ? info->subsampling.y[c] : sy_max;
    }
    for (int c = 0; c < nc; ++c) {
        if (!src->image[c] || !dst->image[c] || dst->pitch[c] < width || src->pitch[c] < info->sizes_x[c])
            return JPEGGPU_INVALID_ARGUMENT;
        const hipError_t err = jg::launch_upsample(
            src->image[c], src->pitch[c], info->sizes_x[c], info->sizes_y[c],
            dst->image[c], dst->pitch[c], width, height,
            info->subsampling.x[c], sx_max, info->subsampling.y[c], sy_max, stream);
        if (err != hipSuccess) return JPEGGPU_INTERNAL_ERROR;
    }
    return JPEGGPU_SUCCESS;
}

enum jpeggpu_status jpeggpu_ext_planes_to_rgbi(
    const struct jpeggpu_img_info* info,
    const struct jpeggpu_img* src,
    uint8_t* dst,
    int dst_pitch,
    int width,
    int height,
    jpeggpu_stream_t stream)
{
    if (!info || !src || !dst || width <= 0 || height <= 0 || dst_pitch < 3 * width) return JPEGGPU_INVALID_ARGUMENT;
    const int nc = info->num_components;
    if (nc != 1 && nc != 3) return JPEGGPU_NOT_SUPPORTED; // as the reference's helper (util/util.h:42-45)
    int sx_max = 0, sy_max = 0;
    for (int c = 0; c < nc; ++c) {
        if (info->subsampling.x[c] < 1 || info->subsampling.y[c] < 1) return JPEGGPU_INVALID_ARGUMENT;
        if (!src->image[c] || src->pitch[c] < info->sizes_x[c]) return JPEGGPU_INVALID_ARGUMENT;
        sx_max = info->subsampling.x[c] > sx_max ? info->subsampling.x[c] : sx_max;
        sy_max = info->subsampling.y[c] > sy_max ? info->subsampling.y[c] : sy_max;
    }
    const hipError_t err = jg::launch_rgbi(
        src->image, src->pitch, info->sizes_x, info->sizes_y, info->subsampling.x, info->subsampling.y,
        sx_max, sy_max, nc, dst, dst_pitch, width, height, stream);
    return err == hipSuccess ? JPEGGPU_SUCCESS : JPEGGPU_INTERNAL_ERROR;
}

enum jpeggpu_status jpeggpu_ext_self_test(jpeggpu_stream_t stream)
{
    // One small decode through the public calls, planes hashed against constants. The write pass refills its bit window
    // with a counted wait the compiler knows nothing about (jg_kernels.hip, RowWindow; the build checks the generated code,
    // jpeggpu_amd/build.py): this is the same question asked of the running system -- driver, firmware, device.
    jpeggpu_decoder_t dec = nullptr;
    if (jpeggpu_decoder_startup(&dec) != JPEGGPU_SUCCESS) return JPEGGPU_OUT_OF_HOST_MEMORY;
    jpeggpu_status result = JPEGGPU_INTERNAL_ERROR;
    void* d_tmp           = nullptr;
    uint8_t* d_planes     = nullptr;
    std::vector<uint8_t> host;
    do {
        jpeggpu_img_info info;
        if (jpeggpu_decoder_parse_header(dec, &info, jg::kSelfTestJpeg, sizeof(jg::kSelfTestJpeg)) != JPEGGPU_SUCCESS) break;
        size_t tmp_size = 0, total = 0, off[3];
        if (jpeggpu_decoder_get_buffer_size(dec, &tmp_size) != JPEGGPU_SUCCESS || info.num_components != 3) break;
        for (int c = 0; c < 3; ++c) {
            if (info.sizes_x[c] != jg::kSelfTestW[c] || info.sizes_y[c] != jg::kSelfTestH[c]) break;
            off[c] = total;
            total += static_cast<size_t>(jg::kSelfTestW[c]) * jg::kSelfTestH[c];
        }
        if (total != 96u * 80u + 2u * 48u * 40u) break;
        if (hipMalloc(&d_tmp, tmp_size) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_planes), total) != hipSuccess) break;
        jpeggpu_img img{};
        for (int c = 0; c < 3; ++c) {
            img.image[c] = d_planes + off[c];
            img.pitch[c] = jg::kSelfTestW[c];
        }
        if (jpeggpu_decoder_transfer(dec, d_tmp, tmp_size, stream) != JPEGGPU_SUCCESS) break;
        if (jpeggpu_decoder_decode(dec, &img, d_tmp, tmp_size, stream) != JPEGGPU_SUCCESS) break;
        try {
            host.resize(total);
        } catch (const std::bad_alloc&) {
            result = JPEGGPU_OUT_OF_HOST_MEMORY;
            break;
        }
        if (hipMemcpyAsync(host.data(), d_planes, total, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess)
            break;
        bool same = true;
        for (int c = 0; c < 3; ++c) {
            uint64_t h = 0xcbf29ce484222325ull; // FNV-1a
            for (size_t i = 0, n = static_cast<size_t>(jg::kSelfTestW[c]) * jg::kSelfTestH[c]; i < n; ++i) h = (h ^ host[off[c] + i]) * 0x100000001b3ull;
            same = same && h == jg::kSelfTestHash[c];
        }
        result = same ? JPEGGPU_SUCCESS : JPEGGPU_INTERNAL_ERROR;
    } while (false);
    (void)hipGetLastError();
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_planes) (void)hipFree(d_planes);
    (void)jpeggpu_decoder_cleanup(dec);
    return result;
}

enum jpeggpu_status jpeggpu_ext_parse_headers(
    const struct jpeggpu_ext_parse_item* items, int num_items, int num_threads, enum jpeggpu_status* statuses)
{
    if (!items || num_items < 0 || !statuses) return JPEGGPU_INVALID_ARGUMENT;
    for (int i = 0; i < num_items; ++i) {
        statuses[i] = JPEGGPU_INVALID_ARGUMENT;
        for (int j = 0; j < i; ++j) // one decoder must not be used concurrently
            if (items[i].decoder && items[i].decoder == items[j].decoder) return JPEGGPU_INVALID_ARGUMENT;
    }
    if (num_threads < 1) num_threads = 1;
    if (num_threads > num_items) num_threads = num_items;
    std::atomic<int> next{0};
    const auto work = [&]() {
        for (int i = next.fetch_add(1); i < num_items; i = next.fetch_add(1)) {
            const jpeggpu_ext_parse_item& it = items[i];
            statuses[i] = jpeggpu_decoder_parse_header(it.decoder, it.img_info, it.data, it.size);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < num_threads; ++t) pool.emplace_back(work);
    } catch (...) {
        // fewer threads than asked for: the calling thread picks up the rest
    }
    work();
    for (std::thread& t : pool) t.join();
    for (int i = 0; i < num_items; ++i)
        if (statuses[i] != JPEGGPU_SUCCESS) return statuses[i];
    return JPEGGPU_SUCCESS;
}

} // extern "C"
