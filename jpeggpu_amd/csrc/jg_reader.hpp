// jg_reader.hpp -- host-side marker/segment parser (C++, runs on the calling thread, no GPU work).
//
// Behavioural counterpart of the reference's reader (src/reader.hpp:38-196, src/reader.cpp:81-729):
// SOI / SOF0 / SOF1 / DHT / DQT / DRI / SOS / EOI, the same status codes for the same defects, the
// same geometry rules (plane size ceil(size*ss/ss_max), MCU-rounded data size, single-component
// frames forced to 1x1 sampling). Differences, each deliberate:
//   B-1 Huffman tables persist across scans as T.81 requires (the reference loses them);
//   B-2 a non-interleaved scan has one data unit per MCU and ceil(size/8) blocks per row;
//   B-6 length fields and table ids are range-checked;
//   fill bytes (FF FF .. before a marker) are skipped; a scan whose restart-marker count does not
//   match the frame geometry is rejected (the kernels index output by segment number);
//   the walk over the entropy-coded bytes also emits the destuff work list (see jg_defs.h).
#ifndef JG_READER_HPP_
#define JG_READER_HPP_

#include "jg_defs.h"

#include <jpeggpu/jpeggpu.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

namespace jg {

struct Logger {
    bool enabled = false;
    void log(const char* fmt, ...) const __attribute__((format(printf, 2, 3)))
    {
        if (!enabled) return;
        va_list ap;
        va_start(ap, fmt);
        vprintf(fmt, ap);
        va_end(ap);
    }
};

struct Component {
    uint8_t id;
    uint8_t qidx;
    int hs, vs;         // sampling factors (1x1 forced for single-component frames)
    int size_x, size_y; // plane size
};

struct ScanComponent {
    int comp_idx;
    int dc_id, ac_id;     // Huffman table ids 0..3
    int h, v;             // data units per MCU in this scan (1,1 when not interleaved)
    int data_x, data_y;   // plane size rounded up to this scan's MCU
};

struct Scan {
    int num_comp = 0;
    ScanComponent comp[kMaxComp];
    size_t begin = 0; // file offset of first entropy-coded byte
    size_t end   = 0; // file offset of the marker that terminates the scan
    int du_per_mcu = 0;
    int mcus_x = 0, mcus_y = 0;
    int mcus_per_segment = 0;
    int num_subseq = 0;
    int num_du = 0;
    // Segment shard (jpeggpu_ext_set_segment_shard): the tables below describe only this decoder's share of the
    // scan's restart segments, renumbered from 0; data unit 0 of the share lies in MCU `first_mcu` of the frame and
    // the share holds `shard_mcus` MCUs (0: no shard, the whole scan).
    int first_mcu  = 0;
    int shard_mcus = 0;
    int first_segment = 0, total_segments = 0;
    std::vector<uint8_t> table_pack;      // tables in force at SOS that this scan's components select (write pass)
    std::vector<uint8_t> table_pack_sync; // the same tables in the form of the state-only passes (jg_defs.h)
    uint32_t cursor_off = 0, cursor_off_sync = 0; // byte offset of the cursor ring in each pack
    std::vector<Segment> segments;
    std::vector<DestuffChunk> chunks;
    std::vector<int> tail_parts; // subsequence ranges [parts[i], parts[i+1]) cut at segment starts
    // Device-side front end (jg_front.hip): the host does not walk the scan; num_subseq and the three
    // capacities are upper bounds from the header, segments / chunks / tail_parts stay empty.
    bool device_walk    = false;
    uint32_t front_win0 = 0; // device_walk: first 4 KiB window of the transferred bytes the device looks at (the one that holds `begin`)
    int expect_segments = 0;
    int max_chunks      = 0;
    int max_tail_parts  = 0; // entries of the device's tail_parts array
};

struct Stream {
    int size_x = 0, size_y = 0;
    int hs_max = 0, vs_max = 0;
    int num_comp = 0;
    Component comp[kMaxComp];
    int restart_interval = 0;
    int num_scans = 0;
    Scan scans[kMaxScans];
    uint16_t qtable[4][64]; // natural order (reference src/defs.hpp:87-89 keeps 8 bits; 16-bit entries are read too)
    // Transferred byte range of the file: [xfer_begin, xfer_end). Buffer offset = file offset - xfer_begin.
    size_t xfer_begin = 0;
    size_t xfer_end   = 0;
};

/// Requests for a per-image choice (Reader::parse): 0 or -N, N = about how many images share the call (kBatchHintFull:
/// "a batch", jpeggpu_ext_set_batched).
constexpr int kBatchHintFull = 64;
constexpr int kSubseqAutoLone = 0, kSubseqAutoBatched = -kBatchHintFull;
/// Calls of so few images get the plan of a lone decode (multi-hypothesis tables, the lone decode's subsequence size) and
/// jpeggpu_ext_decode_batch decodes them one by one: the chip is empty either way, and what such a call waits for is the
/// chain of dependent passes the speculation shortens (jg_defs.h).
#ifndef JG_LONE_PLAN_IMAGES
#define JG_LONE_PLAN_IMAGES 1
#endif
constexpr int kLonePlanImages = JG_LONE_PLAN_IMAGES;
inline bool lone_plan(int images_per_call) { return images_per_call <= kLonePlanImages; }

/// Subsequence size for an image whose first scan has `scan_bytes_bound` bytes at most, in `segments` restart segments,
/// decoded in calls of about `images_per_call` such images (0: on its own).
///   * One image at a time: what such a decode waits for is a handful of dependent passes over a subsequence (the
///     multi-hypothesis speculation and the one or two flow passes behind it, jg_defs.h), so shorter subsequences are
///     shorter passes -- as long as the extra lanes still fit the chip side by side and data units are not longer than
///     subsequences (then candidates stop meeting and flows get long: the reference's photo takes 0.81 ms at 32 bytes,
///     0.47 at 64). Measured (tools/probe/latency_by_size.py, p50 at 32 / 64 / 128 bytes): 0.08 MP 0.18 / 0.22 / 0.27 ms,
///     2 MP 0.23 / 0.28 / 0.40, 12 MP 0.44 / 0.41 / 0.49: 32 bytes for scans below 1 MB that have restart segments,
///     64 otherwise (without restart segments there is no multi-hypothesis table, and 64 was best at every size).
///   * Images that share launches and fill the chip: every subsequence costs a fixed amount of state, table loads and
///     scan work, so the longest size wins -- 256 bytes -- unless the restart segments are so short that padding each of
///     them to whole subsequences would be a visible share of the decode (kept below 1/16: a segment is padded by half
///     a subsequence on average), or the scan so small that it would not fill two sequences.
///   * Calls that do NOT fill the chip (round 5; up to about a dozen 12 MP images): the write pass's workgroups live as
///     long as ONE lane needs for its subsequence, whatever the chip holds, and the flow iterations are a chain of
///     whole-subsequence decodes: 128 bytes. Device us per call of 2 / 4 / 8 / 16 cfg-2 images with every flow kept in the
///     sequence kernel (tools/probe/batch_curve.py): 64 bytes 393 / 478 / 696 / 1163, 128 bytes 427 / 469 / 568 / 927,
///     256 bytes 530 / 561 / 626 / 823; the reference's photo 2 / 8: 634 / 1060, 603 / 741, 712 / 834.
constexpr size_t kSmallCallBytes = 36u << 20; // scan bytes of a call below which it counts as small
inline int choose_subseq_bytes(int images_per_call, size_t scan_bytes_bound, size_t segments)
{
    if (segments == 0) segments = 1;
    const size_t per_segment = scan_bytes_bound / segments;
    const bool batched = !lone_plan(images_per_call);
    int b = batched ? 256 : 64;
    if (batched && static_cast<size_t>(images_per_call) * scan_bytes_bound < kSmallCallBytes) b = 128;
    if (!batched && segments > 1 && scan_bytes_bound < (1u << 20)) b = 32;
    while (b > 32 && per_segment < static_cast<size_t>(8 * b)) b >>= 1;
    while (b > 64 && scan_bytes_bound < static_cast<size_t>(2 * kSeqSubseq) * static_cast<size_t>(b)) b >>= 1;
    return b;
}

struct Reader {
    /// `device_scan`: the LAST scan of a file -- the one that completes the frame's components; the only one of most
    /// files -- is not walked on the host (jg_front.hip does it on the device); parsing stops at that scan's first
    /// entropy-coded byte. Earlier scans are walked on the host: the next scan header lies behind their last byte; and
    /// the last of several is handed to the device only if it holds at least as many bytes as those in front of it.
    /// `subseq_bytes`: 32, 64, 128 or 256, or a request to choose per image (the reference leaves this as a TODO,
    /// src/decoder_defs.hpp:28-34): kSubseqAutoLone for an image decoded on its own (jpeggpu_decoder_decode),
    /// kSubseqAutoBatched for one that shares its launches with others (jpeggpu_ext_decode_batch). The choice is made
    /// at the first scan header from the bytes left in the file and the restart density (choose_subseq_bytes) and
    /// holds for every scan of the image; subseq_bytes() says what it was.
    jpeggpu_status parse(const uint8_t* data, size_t size, int subseq_bytes, const Logger& log, bool device_scan = false,
                         int shard_rank = 0, int shard_world = 1);
    int subseq_bytes() const { return subseq_bytes_; }

    Stream s;

  private:
    const uint8_t* base_ = nullptr;
    const uint8_t* cur_  = nullptr;
    const uint8_t* end_  = nullptr;
    int subseq_bytes_    = 128;
    int subseq_request_  = 128; // what parse was asked for (a size or one of the kSubseqAuto values)
    bool found_sof_      = false;
    bool qt_defined_[4]{};
    bool dc_defined_[4]{}, ac_defined_[4]{};
    std::vector<uint8_t> dc_tab_[4], ac_tab_[4];           // device-format tables by table id (T.81 Th)
    std::vector<uint8_t> dc_tab_sync_[4], ac_tab_sync_[4]; // their sync-pack form
    std::vector<uint8_t> dht_key_[2][4];                   // DHT payload each table was built from (class, id)
    bool comp_in_scan_[kMaxComp]{};
    bool device_scan_ = false;
    bool stop_        = false; // device mode: nothing behind the scan header is parsed on the host

    size_t remaining() const { return static_cast<size_t>(end_ - cur_); }
    uint8_t u8() { return *cur_++; }
    uint16_t u16()
    {
        const uint16_t hi = u8();
        return static_cast<uint16_t>(hi << 8 | u8());
    }
    jpeggpu_status read_sof(const Logger& log);
    jpeggpu_status read_dht(const Logger& log);
    jpeggpu_status read_dqt(const Logger& log);
    jpeggpu_status read_dri(const Logger& log);
    jpeggpu_status read_sos(const Logger& log);
    jpeggpu_status walk_scan(Scan& scan, const Logger& log);
    jpeggpu_status skip_segment(const Logger& log);
    jpeggpu_status apply_segment_shard(int rank, int world, const Logger& log);
};

/// Build the device form of one Huffman table from a DHT payload
/// (reference compute_huffman_table, src/reader.cpp:186-224).
void build_huff_table(
    std::vector<uint8_t>& t, const uint8_t (&num_codes)[16], const uint8_t* huffval, int count, bool is_dc);
/// The sync-pack form of a table built by build_huff_table: 32-bit first-level entries with multi-symbol high
/// halves (jg_defs.h), everything behind the first level unchanged.
void widen_huff_table(const std::vector<uint8_t>& t, bool is_dc, std::vector<uint8_t>& out);

} // namespace jg

#endif // JG_READER_HPP_
