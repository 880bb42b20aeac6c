/*
 * jpeggpu_ext.h -- additive entry points next to the drop-in API of jpeggpu.h. Nothing here exists in
 * the reference; callers that only use jpeggpu.h never need this header.
 *
 *   jpeggpu_ext_set_subsequence_bytes  tuning knob the reference leaves as a compile-time constant
 *   jpeggpu_ext_set_batched            (src/decoder_defs.hpp:28-34 `chunk_size`); chosen per image by default
 *   jpeggpu_ext_get_layout             where the intermediate buffers of the last parsed image sit
 *   jpeggpu_ext_set_segment_shard      decode a share of one image's restart segments (one image over several GPUs)
 *                                      inside d_tmp, for stage-level parity tests and profiling
 *   jpeggpu_ext_set_profiling /        per-stage device time of a decode from HIP events recorded on the
 *   jpeggpu_ext_get_stage_ms           caller's stream (the reference has wall-clock timing only,
 *                                      benchmark/benchmark_jpeggpu.hpp:96-102)
 *   jpeggpu_ext_decode_batch           one launch per stage for many images; the
 *                                      reference decodes one image per call sequence
 *   jpeggpu_ext_set_device_scan        restart-marker scan and segment / work-list construction on the device
 *   jpeggpu_ext_parse_headers          parse_header of many images on a pool of host threads
 *   jpeggpu_ext_planes_to_rgbi         chroma replication + YCbCr -> interleaved RGB8 (util/util.h:62-104)
 *   jpeggpu_ext_upsample_planes        nearest-neighbour chroma replication on the device, the integer
 *                                      part of the reference's host helper util/util.h:62-91
 */
#ifndef JPEGGPU_JPEGGPU_EXT_H_
#define JPEGGPU_JPEGGPU_EXT_H_

#include <jpeggpu/jpeggpu.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Subsequence size (the reference's compile-time `chunk_size`, 128, with the TODO "pick per image",
 * src/decoder_defs.hpp:28-34). By default the library picks it PER IMAGE at jpeggpu_decoder_parse_header from the size
 * of the scan, its restart density and the call type: 64 bytes for an image decoded on its own (the sequence kernel's
 * serial chain is what such a decode waits for), 256 for one that shares its launches with others, less where restart
 * segments are so short that padding them to whole subsequences would show (jpeggpu_ext_layout.subsequence_bytes says
 * what an image got). jpeggpu_ext_set_batched tells the decoder which call type its images are for;
 * jpeggpu_ext_set_subsequence_bytes fixes the size instead (32, 64, 128 or 256; 0 = back to the per-image choice), as
 * does the environment variable JPEGGPU_SUBSEQ_BYTES at startup. Both take effect at the next parse_header.
 * jpeggpu_ext_decode_batch accepts any mix of sizes (one group of launches per size). */
enum jpeggpu_status jpeggpu_ext_set_subsequence_bytes(jpeggpu_decoder_t decoder, int subseq_bytes);
enum jpeggpu_status jpeggpu_ext_set_batched(jpeggpu_decoder_t decoder, int batched);
/* The plan knows the batch size: tell the decoder ABOUT how many images of this kind share one jpeggpu_ext_decode_batch
 * call (0: decoded on its own, the default; jpeggpu_ext_set_batched(decoder, 1) is the hint 64, "the chip is full"). What
 * is chosen at the next parse_header -- subsequence size, multi-hypothesis tables -- is then made for a launch of that
 * size, and jpeggpu_ext_decode_batch picks its kernels from the subsequences a call REALLY holds. A wrong hint costs speed,
 * never correctness: any decoder may be passed to either decode call. */
enum jpeggpu_status jpeggpu_ext_set_batch_hint(jpeggpu_decoder_t decoder, int images_per_call);

struct jpeggpu_ext_scan_layout {
    int num_components;        /* components in this scan */
    int component_idx[JPEGGPU_MAX_COMP];
    int num_subsequences;
    int num_segments;
    int num_sequences;         /* workgroups of the Huffman kernels */
    int num_data_units;
    int data_units_per_mcu;
    int num_chunks;            /* destuff work items */
    /* byte offsets inside d_tmp */
    size_t off_segments;       /* {int subseq_offset, subseq_count}[num_segments] */
    size_t off_chunks;
    size_t off_destuffed;      /* destuffed bytes, TILED in rows of W + 3 words (W = subsequence_bytes / 4): slot s of
                                  subsequence t is 32-bit word ((t / R) * (W + 3) + s) * R + t % R, R = 32 rows per
                                  tile (16 for 256-byte subsequences); slots 1..W are
                                  the subsequence's own words, slot 0 mirrors the last word of subsequence t - 1,
                                  slots W + 1 and W + 2 the first two of t + 1; a word holds its four stream bytes
                                  most significant first (stream byte i is byte 3 - i % 4) */
    size_t off_segment_index;  /* int[num_subsequences] */
    size_t off_state_p;        /* int[num_subsequences] */
    size_t off_state_n;
    size_t off_state_cz;       /* c | z << 8 */
    size_t off_state_dc01;     /* uint32[num_subsequences]: wrapping 16-bit DC-difference sums of scan
                                  components 0 (low half) and 1 (high half) */
    size_t off_state_dc23;     /* same for scan components 2 and 3 */
    size_t off_symbols;        /* uint16 entries, contiguous per data unit: the unit's DC value (absolute) first, then
                                  one entry per non-zero AC coefficient, value << 6 | zig-zag index (value in
                                  -512..511); a coefficient of magnitude 512 or more (category >= 10) is followed by an escape entry with
                                  index 0 holding value >> 10 in its high bits. Logically one region of symbol_region_entries per
                                  subsequence; physically the regions of 64 consecutive subsequences are interleaved
                                  in sectors of 16 entries: sector j of subsequence s starts at entry
                                  ((s / 64) * (symbol_region_entries / 16) + j) * 1024 + (s % 64) * 16 */
    size_t off_du_table;       /* {uint32 physical index of the first entry, uint32 count (| 128 if the unit holds an
                                  escape)}[num_data_units], stream order; entry k of a unit: w = (first & 15) + k -> (first & ~15) + (w >> 4) * 1024 + (w & 15) */
    int symbol_region_entries;
    /* jpeggpu_ext_set_device_scan: num_subsequences / num_segments / num_chunks above are then capacities from the
     * header; the real counts are uint32 words 1.. at off_device_status (status, subsequences, segments, chunks,
     * tail parts), and off_segments / off_chunks point at tables the device has built. */
    int device_scan;
    size_t off_device_status;
    /* Candidates per subsequence of the multi-hypothesis speculation that a lone decode (jpeggpu_decoder_decode) of this
     * scan runs in front of its synchronisation -- one per data unit of the MCU --, 0 where it does not apply (one data
     * unit per MCU, a decoder for batches, a device-scanned image without restart markers, a scan of more than 256 blocks)
     * or JPEGGPU_MULTI_HYPOTHESIS=0 switched it off at startup. */
    int hypotheses;
    /* Blocks of the block-wise chain walk: 0 where every restart segment has at most 1024 subsequences (one workgroup
     * walks a segment), else the scan's segments cut into blocks of 1024 (a scan without restart markers is one segment). */
    int hypothesis_blocks;
};

struct jpeggpu_ext_layout {
    int subsequence_bytes;
    int subsequences_per_sequence; /* owned by one workgroup of the Huffman kernels */
    int num_scans;
    size_t transferred_bytes;  /* entropy-coded byte range copied by jpeggpu_decoder_transfer */
    size_t blob_bytes;         /* table blob copied by jpeggpu_decoder_transfer */
    size_t off_bytes;          /* stuffed bytes inside d_tmp */
    size_t off_qtables;        /* uint16[4][64], natural order */
    struct jpeggpu_ext_scan_layout scans[JPEGGPU_MAX_COMP];
    int shard_rank, shard_world; /* jpeggpu_ext_set_segment_shard as it applies to this image (0, 1: the whole image) */
};

enum jpeggpu_status jpeggpu_ext_get_layout(jpeggpu_decoder_t decoder, struct jpeggpu_ext_layout* layout);

/* Restart-interval sharding of ONE image (one large image over the GPUs of a node): after
 * jpeggpu_ext_set_segment_shard(decoder, rank, world) -- before parse_header -- the decoder transfers and decodes only
 * restart segments [rank * n / world, (rank + 1) * n / world) of the n the scan has, and writes only the rows of each
 * plane that those segments cover (jpeggpu_ext_get_shard_rows, after parse_header); `world` decoders, on `world`
 * devices or one, write disjoint bands that together are the image. Segments are independent for the Huffman decode
 * and for DC prediction (reference src/decode_dc.cu:119-144). parse_header returns JPEGGPU_NOT_SUPPORTED unless the
 * file has one scan with a restart interval of whole MCU rows; the host walk is used (no device scan).
 * world = 1 switches it off. */
enum jpeggpu_status jpeggpu_ext_set_segment_shard(jpeggpu_decoder_t decoder, int rank, int world);
enum jpeggpu_status jpeggpu_ext_get_shard_rows(jpeggpu_decoder_t decoder, int component, int* first_row, int* num_rows);

/* Device-side front end (enable != 0, before parse_header): parse_header stops at the header of the file's LAST scan
 * (the only one of most files) instead of walking the entropy-coded bytes for restart markers (the
 * reference does that walk on the host inside its timed loop, src/reader.cpp:447-489; here it is 0.2 of the
 * 0.87 ms of a 12 MP image). transfer then copies everything up to the end of the file, and decode first runs four
 * small kernels that find the markers and build the segment table and the destuff work list in device memory. What
 * the host walk reports at parse time -- a scan without terminating marker, a restart-marker count that does not
 * match the geometry, an FF FF 00 sequence: JPEGGPU_INVALID_JPEG in each case, from either walk -- is then known
 * only on the device: decode leaves the planes untouched, and jpeggpu_ext_get_device_status (which synchronises
 * `stream`) returns the status. If the file ends in an end-of-image marker, or in one followed by padding, the
 * copy stops there (a backwards search on the host); otherwise it runs to the end of the file.
 * In a file of several scans the LAST one is the device's if it holds at least as many bytes as the scans in front of it
 * (those are walked on the host: the next scan header lies behind their last byte); otherwise the host walks them all. A batch may mix both
 * kinds: the front end of its device-scanned images runs as four launches for the whole batch (grid.y = image). */
/* enable: 0 off (default); 1 on, the caller asks for the status as above; 2 on and CHECKED: jpeggpu_decoder_decode
 * itself waits for the stream and returns the device's status (it then blocks the host, unlike every other mode).
 * The environment variable JPEGGPU_DEVICE_SCAN switches the scan on at jpeggpu_decoder_startup for callers of the
 * drop-in API alone: "1", "2" or "checked" select the CHECKED mode -- such a caller cannot ask for the device's verdict,
 * so decode tells it, at the price of a blocking call --, "async" mode 1 (a truncated scan then shows as unwritten planes
 * only). Items of jpeggpu_ext_decode_batch are never waited
 * for: their status is read with the call below. A decoder in segment-shard mode (jpeggpu_ext_set_segment_shard with
 * world > 1) always takes the host walk -- its share is cut out of the host walk's tables -- and parse_header logs
 * that the device scan was not used (jpeggpu_ext_layout.scans[0].device_scan says which walk an image got). */
enum jpeggpu_status jpeggpu_ext_set_device_scan(jpeggpu_decoder_t decoder, int enable);
enum jpeggpu_status jpeggpu_ext_get_device_status(
    jpeggpu_decoder_t decoder, const void* d_tmp, jpeggpu_stream_t stream, enum jpeggpu_status* status);

/* Stage timing: when enabled, jpeggpu_decoder_decode records HIP events on the caller's stream
 * between its launches; after the stream has been synchronised jpeggpu_ext_get_stage_ms returns the
 * mean milliseconds per stage (summed over scans) of the decodes since the previous call (at most the
 * last 64), and starts a new measurement window. */
enum jpeggpu_ext_stage {
    JPEGGPU_EXT_STAGE_FRONT      = 0, /* device-side marker scan (jpeggpu_ext_set_device_scan), else ~0 */
    JPEGGPU_EXT_STAGE_DESTUFF    = 1,
    JPEGGPU_EXT_STAGE_SYNC_INTRA = 2,
    JPEGGPU_EXT_STAGE_SYNC_INTER = 3,
    JPEGGPU_EXT_STAGE_TAILS      = 4,
    JPEGGPU_EXT_STAGE_WRITE      = 5,
    JPEGGPU_EXT_STAGE_IDCT       = 6,
    JPEGGPU_EXT_NUM_STAGES       = 7
};
enum jpeggpu_status jpeggpu_ext_set_profiling(jpeggpu_decoder_t decoder, int enable);
enum jpeggpu_status jpeggpu_ext_get_stage_ms(jpeggpu_decoder_t decoder, float* ms /* [JPEGGPU_EXT_NUM_STAGES] */);

/* Batched decode: ONE launch per stage for all scans of all items (grid.y = scan), which is what fills
 * a 256-CU device; the drop-in jpeggpu_decoder_decode launches per image. Every item must have been
 * parsed and transferred (jpeggpu_decoder_transfer) into its own d_tmp; items whose images got different
 * subsequence sizes are launched as one group per size. `d_scratch` is caller-owned device memory of at least
 * jpeggpu_ext_batch_scratch_size(total number of scans) bytes (job descriptors and front-end parameters),
 * private to the stream. The batch handle owns page-locked host staging only: a ring of FOUR staging buffers for
 * the job descriptors. What a call launches follows from what it holds (round 5): a call of ONE image whose decoder was
 * planned for lone decodes (the default, or jpeggpu_ext_set_batch_hint(decoder, 0 or 1)) is decoded as
 * jpeggpu_decoder_decode would decode it -- multi-hypothesis speculation included, never blocking --; a call of fewer
 * than 220 000 subsequences (about nineteen 12 MP images at 256 bytes) keeps every synchronisation flow in its sequence
 * kernel, as a lone decode does; a larger one runs one flow iteration there and the rest in the tail kernel.
 * Like jpeggpu_decoder_decode the call only enqueues -- with one exception: the fifth call in
 * a row on one handle waits (hipEventSynchronize) until the copy of the first one has executed, i.e. the host can
 * run at most four batch calls ahead of the device per handle. */
struct jpeggpu_batch;
typedef struct jpeggpu_batch* jpeggpu_batch_t;
struct jpeggpu_ext_batch_item {
    jpeggpu_decoder_t decoder;
    struct jpeggpu_img* img;
    void* d_tmp;
    size_t tmp_size;
};
size_t jpeggpu_ext_batch_scratch_size(int max_scans);
enum jpeggpu_status jpeggpu_ext_batch_create(jpeggpu_batch_t* batch, int max_scans);
enum jpeggpu_status jpeggpu_ext_decode_batch(
    jpeggpu_batch_t batch,
    const struct jpeggpu_ext_batch_item* items,
    int num_items,
    void* d_scratch,
    size_t scratch_size,
    jpeggpu_stream_t stream);
enum jpeggpu_status jpeggpu_ext_batch_destroy(jpeggpu_batch_t batch);
/* Lock-step flow iterations inside the per-sequence sync kernel (>= 1: the first one is what gives every
 * subsequence its coefficient count and DC sums) before unfinished flows are handed to the low-footprint,
 * re-packing tail kernel (default 1: measured, a second iteration inside the sequence kernel costs a batch more than
 * the tail kernel's trip it saves; the drop-in decode keeps all flows in the sequence kernel, re-packed into its
 * lowest lanes between iterations). */
enum jpeggpu_status jpeggpu_ext_batch_set_sync_iterations(jpeggpu_batch_t batch, int iterations);
/* For a caller that uses ONE stream: split every batch into `parts` (1..4, default 1) that run concurrently,
 * part 0 on the caller's stream and the others on internal streams forked from and joined back into it with
 * events, so that one part's latency-bound synchronisation tail overlaps another part's decode (+15 % with
 * 2-3 parts). A caller that already keeps several streams busy gains nothing. Stage timing then reports
 * part 0. */
enum jpeggpu_status jpeggpu_ext_batch_set_overlap(jpeggpu_batch_t batch, int parts);
/* A call that fills the chip (what jpeggpu_ext_decode_batch launches is described above) runs the parts of the
 * synchronisation's tail and the sequences of the write pass as ONE launch, huff_tail_write: the sequences of parts
 * that are done are written while the slow parts still run (default on, or the environment's JPEGGPU_FUSE_TAIL_WRITE=0
 * when the batch is created; off: the two kernels one after the other, as up to round 4). Stage timing then reports
 * the launch under "write" and nothing under "sync_inter". Not taken with a caller's cap of the sequence kernel's
 * iterations (jpeggpu_ext_batch_set_sync_iterations), nor by calls of more than 256 scans. */
enum jpeggpu_status jpeggpu_ext_batch_set_fused_tail(jpeggpu_batch_t batch, int enable);
/* Writers of huff_tail_write that gave up waiting for a sequence to become ready, since the library was loaded (their
 * wait is bounded so that a defect cannot hang the GPU; 0 on every correct run: tests and the soak assert it). */
enum jpeggpu_status jpeggpu_ext_fused_tail_timeouts(unsigned int* count);
/* Stage timing of batched decodes; same contract as jpeggpu_ext_set_profiling / _get_stage_ms. */
enum jpeggpu_status jpeggpu_ext_batch_set_profiling(jpeggpu_batch_t batch, int enable);
enum jpeggpu_status jpeggpu_ext_batch_get_stage_ms(jpeggpu_batch_t batch, float* ms /* [JPEGGPU_EXT_NUM_STAGES] */);

/* Replicate every plane of `src` (as produced by jpeggpu_decoder_decode for `info`) to the full
 * image resolution: dst[c][y][x] = src[c][y * sy_c / sy_max][x * sx_c / sx_max]. */
enum jpeggpu_status jpeggpu_ext_upsample_planes(
    const struct jpeggpu_img_info* info,
    const struct jpeggpu_img* src,
    struct jpeggpu_img* dst,
    int width,
    int height,
    jpeggpu_stream_t stream);

/* One small decode (a built-in 96 x 80 4:2:0 JPEG with restart markers, through the public calls above, device memory
 * from hipMalloc) whose planes are compared with stored hashes: JPEGGPU_SUCCESS if the running system -- library build,
 * driver, device -- decodes bit-exactly, JPEGGPU_INTERNAL_ERROR if not (or without a device). Synchronises `stream`.
 * Meant for an application's start-up checks; the library never calls it on its own. */
enum jpeggpu_status jpeggpu_ext_self_test(jpeggpu_stream_t stream);

/* jpeggpu_decoder_parse_header for many images on `num_threads` host threads (the calling thread is one
 * of them). A 12 MP scan costs ~0.2 ms of one core to walk (reference src/reader.cpp:447-489 does the
 * same walk inside its timed loop), so a serving loop at 18 k images/s needs about four cores of it.
 * Every decoder must appear once. statuses[i] receives the result of item i; the return value is the
 * first failure, or JPEGGPU_SUCCESS. */
struct jpeggpu_ext_parse_item {
    jpeggpu_decoder_t decoder;
    struct jpeggpu_img_info* img_info;
    const uint8_t* data;
    size_t size;
};
enum jpeggpu_status jpeggpu_ext_parse_headers(
    const struct jpeggpu_ext_parse_item* items, int num_items, int num_threads, enum jpeggpu_status* statuses);

/* Planes of a 1- or 3-component image -> interleaved RGB8 at the full image resolution: nearest-neighbour
 * chroma replication + the JFIF YCbCr matrix in float, rounded and clamped -- the arithmetic of the
 * reference's host helper conv_to_rgbi (util/util.h:62-104). dst[y * dst_pitch + 3 * x + {0,1,2}] = R,G,B.
 * JPEGGPU_NOT_SUPPORTED for 2 or 4 components, as the helper. */
enum jpeggpu_status jpeggpu_ext_planes_to_rgbi(
    const struct jpeggpu_img_info* info,
    const struct jpeggpu_img* src,
    uint8_t* dst,
    int dst_pitch,
    int width,
    int height,
    jpeggpu_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* JPEGGPU_JPEGGPU_EXT_H_ */
