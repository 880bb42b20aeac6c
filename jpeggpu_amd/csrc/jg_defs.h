// jg_defs.h -- plain-old-data shared by the host parser, the launch code and the gfx950 kernels.
//
// Vocabulary follows the reference (src/decoder.cpp:46-63): scan, segment (bytes between restart
// markers), subsequence (fixed slice of destuffed segment data decoded by one lane), data unit
// (8x8 block), MCU. A "sequence" here is the run of subsequences owned by one workgroup.
#ifndef JG_DEFS_H_
#define JG_DEFS_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define JG_HD __host__ __device__
#else
#define JG_HD
#endif

namespace jg {

constexpr int kMaxComp      = 4;  // include/jpeggpu/jpeggpu.h:33
constexpr int kMaxScans     = 4;  // baseline: every component appears in exactly one scan
constexpr int kMaxDuPerMcu  = 10; // T.81 B.2.3
constexpr int kSeqLanes    = 256; // lanes per workgroup of the Huffman kernels (reference decode_huffman.cu:777)
constexpr int kSeqOverlap  = 16;  // lanes of the sync kernel that re-decode the tail of the previous sequence
constexpr int kSeqSubseq   = kSeqLanes - kSeqOverlap; // subsequences owned by one workgroup ("sequence") of a LONE decode
// A batch runs ONE flow iteration in the sequence kernel, and the only overlap lane whose work that iteration uses is the
// last one (its flow gives the sequence's first subsequence its n and DC sums): its sequences are 255 subsequences
// long, one lane re-decodes the previous sequence's last subsequence (ScanParams::seq_subseq says which a job has). The
// 15 lanes saved are 6 % of the sequence kernel's and of the write pass's workgroups (round 4).
constexpr int kSeqOverlapBatch = 1;
constexpr int kSeqSubseqBatch  = kSeqLanes - kSeqOverlapBatch;
// Target subsequences per workgroup of huff_sync_tail (cut at segment starts). The kernel is a chain of dependent
// whole-subsequence decodes (two to four trips of its lock-step loop) for the ~8 % of the subsequences whose flow the
// sequence kernel left unfinished, so what it costs is latency, and a part's trip takes as long as the slowest lane of
// its fullest wave. Measured per 64 images of 12 MP, serialized / four streams overlapping (round 4, in-run): parts of
// 2048 in 1024-lane workgroups (round 3) 370 us / 27.4 k images/s; in 256-lane workgroups, a part's flows in its lowest
// lanes: 1024 590 us / 23.7 k (the 1024-lane kernel, two workgroups to a CU), 768 258 / 27.7 k, 512 250 / 27.0 k,
// 384 390 / 26.4 k; with the flows spread over the four waves: 512 237, 640 212, 768 232, 900 223, 960 219, 1260 230,
// 1920 241 us, images/s 27.8 ... 28.7 k, rising slowly with the size.
#ifndef JG_TAIL_PART
#define JG_TAIL_PART 960
#endif
constexpr int kTailPartSubseq = JG_TAIL_PART;
// A batched call of fewer subsequences than this does not fill the chip (256 CUs x 5 workgroups of 255 lanes hold
// 326 000): what it waits for is the chain of dependent flow iterations, as a lone decode does, and its sequence kernel
// keeps every flow in the workgroup (jg_decoder.cpp, decode_batch_impl; jg_kernels.hip, JobArrayLow). Measured per call of
// cfg-2 images at 256 bytes (11 400 subsequences each; us, flows kept / one iteration + marks + tail kernel, round 5):
// 2: 530 / 567, 4: 561 / 613, 8: 626 / 678, 16: 828 / 878, 24: 1194 / 1113, 32: 1502 / 1408, 64: 2833 / 2418; the
// reference's photo 16: 1034 / 1121, 32: 1895 / 1668.
#ifndef JG_KEEP_FLOWS_BELOW
#define JG_KEEP_FLOWS_BELOW 220000
#endif
constexpr long long kKeepFlowsBelowSubseq = JG_KEEP_FLOWS_BELOW;
constexpr int kDestuffWin   = 4096; // stuffed bytes handled by one destuff workgroup (256 lanes x 16 B)

/// Zig-zag index -> raster index inside a data unit (T.81 figure A.6; reference src/defs.hpp:94-102).
#define JG_ORDER_NATURAL                                                                           \
    {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,       \
     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,       \
     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63}

/// Device Huffman table pack. Same information as the reference's `huffman_table`
/// (src/reader.hpp:45-64: 8-bit LUT + maxcode/valptr walk + huffval), re-laid-out for a wave of 64
/// lanes that all walk different bitstreams: a wide first-level LUT of pre-digested 16-bit entries
/// (so that almost no lane ever needs the long-code path, which every lane of the wave would have
/// to wait for), and a long-code path without a dependent chain of table reads.
///
///   one table = lut16[1 << LB] | lim16[8] | valoff16[8] | huffval[256] | sub16[nsub][32]
///                                                                    (LB = 9 for DC, 11 for AC)
///
///   lut16 entry, indexed by the LB most significant bits of the 32-bit window:
///     bits  0..4  total symbol length = code length + magnitude bits, 1..31 (0 = code longer than LB bits)
///     bits  5..8  magnitude category s
///     bits  9..14 zig-zag advance: run + 1 (1..16), or 63 for end-of-block (AC symbol with s == 0 and
///                 run != 15): an advance that reaches index 63 or beyond closes the data unit; 1 in a DC table
///     bit   15    only in the first level of an AC table of the WRITE pack (kEntrySlow): the lane cannot take this
///                 entry in an ordinary step of the write pass's loop -- it has no length (second level / long code) or its
///                 category takes an escape entry -- and waits for the rare slot; read as a signed 16-bit value the entry
///                 is then negative, which is all the loop's step asks (jg_huff_core.h, decode_units)
///   an entry with bits 0..4 == 0 and a non-zero rest is indirect: (entry >> 5) - 1 is the index of a
///     second-level table sub16[k][32], indexed by the next 5 window bits, holding entries of the same
///     format (0 there, or a first-level 0, = take the long-code path below). The host allocates a
///     second-level table for every LB-bit prefix that starts a valid longer code, up to
///     kMaxSubTables per table; with the tables of real files that is a handful (the long codes sit
///     together at the all-ones end of the code space), and no lane then takes the long-code path.
///   lim16[j], j = 0..7: canonical code counter after length 9+j, left-aligned to 16 bits and
///     saturated to 0xFFFF; a 16-bit window v holds a code of length <= 9+j iff v < lim16[j]
///     (equivalent to the reference's maxcode walk, src/decode_huffman.cu:177-187)
///   valoff16[j]: (huffval index of the first code of length 9+j minus that code) mod 256
///
/// A scan's pack holds only the tables its components select, followed by the scan's CURSOR RING:
/// one 16-byte entry per data unit of the MCU,
///   { dc table offset | ac table offset << 16,  16 * scan component | data unit index << 8,
///     byte offset of this entry,  byte offset of the next data unit's entry }   (offsets in the pack)
/// A lane keeps the entry of the data unit it is in; at the end of a data unit it loads the next one.
/// This replaces per-symbol table selection arithmetic by one LDS read per symbol.
constexpr int kLutBitsDc   = 9;
#ifndef JG_LUT_BITS_AC
#define JG_LUT_BITS_AC 11
#endif
constexpr int kLutBitsAc   = JG_LUT_BITS_AC;
constexpr int kSubBits     = 5;
constexpr int kSubTableSize = 2 << kSubBits; // bytes
constexpr int kMaxSubTables = 16;
constexpr int kHuffAuxSize = 16 + 16 + 256;
constexpr int kDcTableSize = (2 << kLutBitsDc) + kHuffAuxSize; // 1312, plus second-level tables
constexpr int kAcTableSize = (2 << kLutBitsAc) + kHuffAuxSize; // 4384, plus second-level tables

/// A scan has TWO table packs. The write pass, which needs every symbol's magnitude, uses the pack described
/// above. The state-only passes (speculation, flows) use the SYNC pack: the same tables with 32-bit first-level
/// entries, low half = the entry above, high half = a MULTI-SYMBOL entry of the same shape for the same window
/// bits:
///     bits 0..4   total length of ALL the symbols it stands for (every code with its magnitude bits lies inside
///                 the LB index bits)
///     bits 5..8   zig-zag advance of all but the last of them (at most 15)
///     bits 9..15  zig-zag advance of all of them (an end-of-block, which can only be the last, counts 63)
/// -- as many AC symbols as fit, decoded with this same table, which is right as long as the data unit does not end
/// in front of the last of them: the symbol loop takes the high half unless index + advance of the earlier symbols
/// reaches 64 (jg_huff_core.h). Where no second symbol fits the high half repeats the low one (its bits 5..8 are
/// then a category; a test that fails for it falls back to the same entry). A state-only pass then takes ~1.6
/// symbols per step on photographic data. DC tables and the second-level tables hold single symbols only.
constexpr int kSyncEntryBytes = 4;
constexpr int kMultiMaxPre    = 15;
constexpr int kDcTableSizeSync = (kSyncEntryBytes << kLutBitsDc) + kHuffAuxSize;
constexpr int kAcTableSizeSync = (kSyncEntryBytes << kLutBitsAc) + kHuffAuxSize;

/// Largest table pack of a scan: four DC and four AC tables with every second-level table, plus the cursor
/// ring. Offsets into the pack are kept in 16 bits (Scan::dc_off / ac_off, the halves of CursorEntry::tabs),
/// and on the device the kernels turn them into absolute LDS addresses that must stay below 64 KiB as well
/// (jg_kernels.hip checks its carve bases against these constants).
constexpr uint32_t kMaxTablePack =
    kMaxComp * (kDcTableSize + kMaxSubTables * kSubTableSize) + kMaxComp * (kAcTableSize + kMaxSubTables * kSubTableSize) + kMaxDuPerMcu * 16;
constexpr uint32_t kMaxTablePackSync =
    kMaxComp * (kDcTableSizeSync + kMaxSubTables * kSubTableSize) + kMaxComp * (kAcTableSizeSync + kMaxSubTables * kSubTableSize) + kMaxDuPerMcu * 16;
static_assert(kMaxTablePack < 65536 && kMaxTablePackSync < 65536, "table offsets are 16-bit");

constexpr uint32_t kEobAdvance = 63u;   // advance field of an end-of-block
constexpr uint32_t kEntrySlow  = 0x8000u;
JG_HD inline uint32_t huff_entry(int codelen, uint32_t sym, bool is_dc)
{
    // DC: sym is the category (hardened to 4 bits; valid baseline streams use 0..11).
    // AC: sym = run << 4 | category (reference decode_huffman.cu:232-259).
    const uint32_t s   = sym & 15u;
    const uint32_t r   = is_dc ? 0u : (sym >> 4);
    const bool eob = !is_dc && s == 0 && r != 15;
    return (static_cast<uint32_t>(codelen) + s) | (s << 5) | ((eob ? kEobAdvance : r + 1u) << 9);
}

/// Layout of the destuffed buffer: TILES of 32 (or 16) subsequences, word-major inside a tile. The lanes
/// of a wave walk 64 different subsequences at about the same pace, so their 4-byte refills fall into a few
/// shared 128-byte lines; with the plain layout every refill touched its own line (lane stride = 128 B) and the
/// write pass fetched 25x the bitstream (rocprofv3 FETCH_SIZE).
///
/// A subsequence's ROW holds its W words (W = words per subsequence) AND copies of its neighbours' words
/// around them: slot 0 = the last word of the previous subsequence, slots 1..W = its own words, slots W+1 and
/// W+2 = the first two words of the next subsequence. A decode of subsequence t starts at most 31 bits in
/// front of it and looks at most 64 bits past its end (32-bit peek + one prefetched word), so every word it
/// touches is in row t and the address of the next word is the previous one plus 128: the refill, which a
/// wave executes in nearly every iteration of the symbol loop (some lane always needs a word), costs one
/// add instead of the seven instructions of the general tiled address. destuff_kernel stores every word up to
/// three times (+9 % bytes at W = 32); rows at the ends of the buffer keep slots nobody wrote, which no
/// committed symbol depends on (jg_kernels.hip, GlobalFetch).
///
///   slot s of row t:  32-bit word ((t / R) * (W + 3) + s) * R + t % R,   R = rows per tile
///
/// R = 32 rows, except 16 at W = 64 (256-byte subsequences): a destuff workgroup's 4 KiB window then still fills
/// WHOLE tiles. With 32 rows of 64 words a window was half a tile, every 128-byte line got its two halves from two
/// workgroups -- usually on two XCDs, i.e. through two L2s -- and destuff_kernel ran 35 % longer than at W = 32.
constexpr int kRowLeadWords = 1; // copies in front of the row's own words
constexpr int kRowTailWords = 2; // copies behind them
constexpr int kRowExtraWords = kRowLeadWords + kRowTailWords;
JG_HD constexpr int tile_rows_log2(int log2_w) { return log2_w >= 6 ? 4 : 5; }
constexpr int kMaxTileRows = 32;
/// 32-bit word index of slot `slot` of row `row`.
JG_HD inline uint32_t tiled_slot(uint32_t row, uint32_t slot, int log2_w)
{
    const int lr = tile_rows_log2(log2_w);
    return (((row >> lr) * ((1u << log2_w) + kRowExtraWords) + slot) << lr) + (row & ((1u << lr) - 1u));
}
/// Word index of the MAIN copy of linear word `linear_word` of the scan.
JG_HD inline uint32_t tiled_word(uint32_t linear_word, int log2_w)
{
    return tiled_slot(linear_word >> log2_w, (linear_word & ((1u << log2_w) - 1u)) + kRowLeadWords, log2_w);
}
/// Bytes of the tiled buffer for `num_subseq` subsequences plus `spare_rows` rows behind them (whole tiles).
JG_HD inline uint64_t tiled_buffer_bytes(uint32_t num_subseq, int subseq_bytes, int spare_rows)
{
    const uint64_t rows = (static_cast<uint64_t>(num_subseq) + static_cast<uint64_t>(spare_rows) + kMaxTileRows - 1) / kMaxTileRows * kMaxTileRows;
    return rows * (static_cast<uint64_t>(subseq_bytes) + 4u * kRowExtraWords);
}

/// The SYMBOL STREAM the write pass emits and the IDCT gathers: 16-bit entries, contiguous per data unit.
///   * a unit's FIRST entry is its DC coefficient, absolute (the predictor is already added), 16 bits;
///   * every other entry is a non-zero AC coefficient: value << 6 | zig-zag index (1..63), value in -512..511 as a
///     10-bit two's complement number in the high bits;
///   * an AC coefficient of magnitude category 10 and above (|value| >= 512, -512 included: quantisers of 1 at best) is followed by an
///     ESCAPE entry, index field 0, whose high bits are value >> 10: value = int16((escape >> 6) << 10 | entry >> 6).
///     The count of a unit that holds an escape has bit 7 set in the data-unit table (a count is at most 127).
/// Half the bytes of a 32-bit entry (index << 16 | value) -- the stream is most of what the write pass stores and the
/// IDCT fetches: with every other sector left out (timing only) the write pass ran 15 % and a batch 11 % faster.
///
/// Most entries a data unit can have: 64 coefficients, every AC one escaped.
constexpr uint32_t kMaxUnitEntries = 64 + 63;

/// Entries reserved per subsequence. A lane of the write pass emits the data units whose DC symbol its subsequence
/// commits. The densest unit a code table allows is 64 entries in 127 bits (a one-bit DC code of category 0, then 63
/// coefficients of one code bit and one magnitude bit each; an escaped coefficient is two entries for twelve bits or
/// more), so the units that start AND end inside the 8 * B bits of the subsequence hold at most ceil(8 * B * 64 / 127)
/// entries, and the unit the lane runs on to finish at most kMaxUnitEntries more. (The round-2 figure, 4 * B + 64 +
/// 16, forgot that the last unit may be all escapes: a crafted table pack overran it by 47 entries at B = 64, and the
/// sink's clamp then dropped entries silently; tests/test_emulation.py holds that file.)
constexpr uint32_t kSymSectorEntries = 16; // entries a lane flushes at a time: one 32-byte sector
JG_HD constexpr uint32_t sym_region_entries(int subseq_bytes)
{
    return ((static_cast<uint32_t>(subseq_bytes) * 8u * 64u + 126u) / 127u + kMaxUnitEntries + kSymSectorEntries - 1u) / kSymSectorEntries * kSymSectorEntries;
}
static_assert(sym_region_entries(32) == 272 && sym_region_entries(64) == 400 && sym_region_entries(128) == 656 && sym_region_entries(256) == 1168,
              "whole sectors, at least ceil(8 B 64 / 127) + kMaxUnitEntries entries");

struct uint2_t {
    uint32_t x, y;
};

/// Placement of the symbol stream in memory. Logically every subsequence has a region of
/// sym_region_entries() entries; physically the regions of 64 consecutive subsequences are interleaved
/// sector by sector: sector j of subsequence s sits at entry ((s / 64) * R + j) * 1024 + (s % 64) * 16, R = sectors
/// per region. The 64 lanes of a wave flush together, so the four sectors of a 128-byte line come from four
/// neighbouring lanes at the same time; with plain regions a line got its sectors from ONE lane over ~32 iterations
/// and was evicted from L2 partially written in between (as for the bitstream: tiled_word above).
constexpr uint32_t kSymTileSubseq = 64;
constexpr uint32_t kSymSectorStride = kSymTileSubseq * kSymSectorEntries; // entries between consecutive sectors of one region
JG_HD inline uint32_t sym_region_base(uint32_t sub, uint32_t region_entries)
{
    return (sub / kSymTileSubseq) * (region_entries / kSymSectorEntries) * kSymSectorStride + (sub % kSymTileSubseq) * kSymSectorEntries;
}
/// Physical index of logical entry `e` of the region that starts at physical `base`.
JG_HD inline uint32_t sym_at(uint32_t base, uint32_t e)
{
    return base + (e / kSymSectorEntries) * kSymSectorStride + (e % kSymSectorEntries);
}
/// Physical index of the k-th entry after the entry at physical index `first` (same region).
JG_HD inline uint32_t sym_advance(uint32_t first, uint32_t k)
{
    const uint32_t w = (first % kSymSectorEntries) + k;
    return (first & ~(kSymSectorEntries - 1u)) + (w / kSymSectorEntries) * kSymSectorStride + (w % kSymSectorEntries);
}
/// Entries of the whole stream buffer, with room for a 128-entry gather starting at any clamped index.
JG_HD inline uint64_t sym_buffer_entries(uint32_t num_subseq, uint32_t region_entries)
{
    const uint64_t tiles = (static_cast<uint64_t>(num_subseq) + kSymTileSubseq - 1) / kSymTileSubseq;
    return tiles * (region_entries / kSymSectorEntries) * kSymSectorStride + 16u * kSymSectorStride;
}
/// One restart segment of a scan inside the destuffed buffer (reference src/reader.hpp:38-43).
struct Segment {
    int subseq_offset; // subsequences before this segment
    int subseq_count;  // subsequences in this segment
};

/// One destuff work item: an aligned 4 KiB window of the transferred bytes intersected with one
/// segment's byte range. The host walk that builds the segment table (reference
/// src/reader.cpp:447-489) already knows how many stuffing bytes precede every position, so the
/// destination offset is handed over instead of being recomputed by device-wide scans.
struct DestuffChunk {
    uint32_t win_off;  // 16-byte aligned offset of the window in the transferred byte buffer
    uint32_t begin;    // first byte of the segment inside the window (buffer offset)
    uint32_t end;      // one past the last byte (buffer offset)
    uint32_t dst_off;  // destination byte offset in the destuffed buffer of the first data byte
    uint32_t pad_end;  // if non-zero: zero-fill the destuffed buffer up to this offset (segment tail)
    int32_t seg;       // segment index
    uint32_t first;    // 1 if `begin` is the first byte of the segment (its predecessor is a marker)
    uint32_t reserved;
};
static_assert(sizeof(DestuffChunk) == 32, "layout is shared with the kernels");

/// Per-scan constants for the Huffman kernels (reference `const_state`, decode_huffman.cu:92-122).
struct ScanParams {
    int num_subseq;
    int num_segments;
    int du_per_mcu;       // data units per MCU in this scan (1 for a non-interleaved scan)
    int num_comp;         // components in this scan
    int mcus_per_segment; // restart interval, or all MCUs when there is none
    int total_mcus;
    int subseq_words;     // 32-bit words per subsequence (subsequence bytes / 4)
    int max_intra_iters;  // lock-step flow iterations inside huff_sync_intra before flows are handed to huff_sync_tail
    uint32_t tab_bytes;   // size of the scan's Huffman table pack (tables + cursor ring) the kernel at hand uses
    uint32_t cursor_off;  // byte offset of the cursor ring in that pack
    uint32_t tab_bytes_sync, cursor_off_sync; // the same for the sync pack (state-only passes)
    int mh;               // hypotheses per subsequence of the multi-hypothesis speculation (below); 0: off
    int seq_subseq;       // subsequences a workgroup of the Huffman kernels owns: kSeqSubseq (lone decode) or kSeqSubseqBatch
    int tail_marks;       // 1: the sequence kernel caps its flow iterations and leaves `pending` marks for huff_sync_tail (a
                          //    full batch); 0: every flow stays in its sequence's workgroup and the tail kernel looks at the
                          //    sequence boundaries only (a lone decode, a small batch). Set beside max_intra_iters (build_jobs).
    /// The state-only kernels call this on their copy of the parameters before loading the tables.
    JG_HD inline void use_sync_pack()
    {
        tab_bytes  = tab_bytes_sync;
        cursor_off = cursor_off_sync;
    }
};

/// MULTI-HYPOTHESIS SPECULATION (one image at a time). The synchronisation of the reference (decode_huffman.cu:413-524)
/// guesses that every subsequence starts a data unit of index 0; a flow then runs until its state meets a stored one,
/// and with several data units per MCU that takes long for one reason: position and zig-zag index of two decoders of the
/// same bits fall into step within a unit or two, the data-unit INDEX never does -- it stays off by a constant until a
/// table mismatch (luma bits read with the chroma tables) throws the decoder out of step and it comes back at another
/// offset. The longest flow of a 12 MP 4:2:0 image passes 17 subsequences of 64 bytes, 32 on the reference's photo, and
/// one image alone waits for exactly that chain. So every subsequence is speculated once per data unit of the MCU
/// (kernel huff_mh_spec: hypothesis h = "a unit of index h starts at my first bit"); a second kernel (huff_mh_flow)
/// decodes, for every candidate exit state, the FOLLOWING subsequence and notes which of ITS candidates it reaches (99 %
/// after one subsequence: one of the candidates has the right index); a third (huff_mh_resolve) walks these links from
/// each restart segment's first subsequence, whose hypothesis 0 is exact, and writes the chain's states as the table
/// huff_sync_intra starts from. That kernel then runs the reference's algorithm unchanged -- its flows verify every
/// entry and supply the coefficient counts -- and meets agreement at once nearly everywhere: 3 passes instead of 17, 6
/// instead of 32. Correctness does not depend on any of this: the flows reach the sequential decoder's states from ANY
/// initial table (tests/test_emulation.py runs it on the host twin).
///   candidates: mh_p / mh_cz [h * num_subseq + sub]; link of a candidate: steps (1..kMhSteps, 0 = none) | the candidate
///   reached << 4 | where the states it passed on the way are kept << 8 (a flow that needs more than one subsequence --
///   2 % do, in detailed regions whose data units are longer than a subsequence -- leaves the exact states of the
///   subsequences it hops over in a pool, mh_pool: kMhSteps - 1 entries {p, c | z << 8} from that index on; the index is
///   kMhNoPool when the pool was full); mh_known[sub]: the table entry of sub is a state of the chain (an entry the
///   chain hopped over without pool entries is filled by the flow from upstream).
constexpr int kMhMaxHyp        = 8;    // 4-bit candidate index; more data units per MCU: plain speculation
#ifndef JG_MH_STEPS
#define JG_MH_STEPS 8
#endif
constexpr int kMhSteps         = JG_MH_STEPS; // subsequences a candidate's flow runs before it gives up
// the step count sits in a 4-bit link field and the pool hands out kMhSteps - 1 entries per flow; a run of entries the
// chain hops over (kMhSteps - 1) must be shorter than the overlap zone of huff_sync_intra, or entries at the start of a
// sequence would get their n / DC sums from no flow
static_assert(kMhSteps >= 2 && kMhSteps <= 15 && kMhSteps - 1 < kSeqOverlap, "JG_MH_STEPS: link field, pool reservation, overlap zone");
constexpr int kMhMaxSegSubseq  = 1024; // the chain walk of a segment -- or of one BLOCK of a longer segment -- happens in LDS
/// Segments longer than that (a scan without restart markers is ONE segment) are walked BLOCK-WISE: every block of up to
/// kMhMaxSegSubseq subsequences first works out, for each of the 64 nodes (candidate h, row 0..7) the chain can enter it
/// at, where the chain leaves it (huff_mh_block_maps: pointer jumping inside the block); one workgroup strings the
/// blocks' maps together (huff_mh_block_chain: block b's entry node = block b - 1's exit for ITS entry node, from node
/// (0, 0) at the segment's start); then every block walks its part of the chain from its entry node as a short segment
/// would (huff_mh_resolve). Three launches instead of one, whatever the segment's length.
constexpr int kMhMaxBlocks     = 256;  // blocks of one scan (their maps are chained in LDS): 16 MB of scan at 64-byte subsequences
constexpr uint32_t kMhBlockBroken = 0xFFFEu; // entry of a block behind the point where the chain broke off
struct MhBlock {
    int first;    // first subsequence of the block
    int count;    // subsequences in it
    int seg_end;  // one past the last subsequence of its segment
    int opens;    // 1: the block starts its segment (entry node (0, 0))
};
constexpr uint32_t kMhNoLink   = 0;
constexpr uint32_t kMhNoPool   = 0xFFFFFFu; // 24-bit pool index
/// Words of ScanJob::fuse_ctl for a scan of `num_seq` sequences (jg_kernels.hip, fuse_init: 16 + 2 * 256 shared ones, then a
/// counter and a queue entry per sequence).
JG_HD constexpr size_t fuse_ctl_words(size_t num_seq) { return 16 + 2 * 256 + 2 * num_seq; }
JG_HD inline uint32_t mh_pool_entries(uint32_t num_subseq) { return 2u * num_subseq + 64u; } // states; one counter in front

struct CursorEntry {
    uint32_t tabs; // dc table offset | ac table offset << 16
    uint32_t meta; // 16 * scan component (shift of its 16-bit DC sum) | data unit index << 8
    uint32_t self; // byte offset of this entry in the pack
    uint32_t next; // byte offset of the next data unit's entry
};
static_assert(sizeof(CursorEntry) == 16, "read with one 16-byte load");

/// Geometry for dequant + IDCT reading the stream-order coefficient buffer (replaces the reference's
/// separate transpose pass, src/decode_transpose.cu:41-132, plus src/idct.cu:146-223).
struct IdctParams {
    int num_du;     // data units in the scan (< 2^31), or in this decoder's share of its restart segments
    int du_per_mcu;
    int mcus_x;
    int first_mcu;  // MCU of the frame that data unit 0 belongs to (non-zero for a segment shard, jpeggpu_ext.h)
    // n / d for n < 2^31 as (mulhi(n, mul) >> shift); mul == 0 stands for d == 1 (jg_defs.h, magic_div)
    uint32_t du_per_mcu_mul, du_per_mcu_shift;
    uint32_t mcus_x_mul, mcus_x_shift;
    uint8_t du_comp[kMaxDuPerMcu]; // scan-component index of each data unit in the MCU
    uint8_t du_dx[kMaxDuPerMcu];   // block column inside the MCU
    uint8_t du_dy[kMaxDuPerMcu];   // block row inside the MCU
    int comp_h[kMaxComp];          // blocks per MCU horizontally (1 when non-interleaved)
    int comp_v[kMaxComp];
    int size_x[kMaxComp];          // visible plane size (crop)
    int size_y[kMaxComp];
    int pitch[kMaxComp];
    int qidx[kMaxComp];            // quantisation table index
    uint8_t* plane[kMaxComp];
};

/// Division by a runtime constant: for 2 <= d and n < 2^31, with L = ceil(log2 d) and
/// M = ceil(2^(31+L) / d) (< 2^32), floor(n / d) == (mulhi(n, M) >> (L - 1)): the error term
/// n * (M * d - 2^(31+L)) / (d * 2^(31+L)) is below 1 / d.
struct MagicDiv {
    uint32_t mul, shift;
};
inline MagicDiv magic_div(uint32_t d)
{
    if (d <= 1) return MagicDiv{0u, 0u};
    uint32_t l = 0;
    while ((1u << l) < d) ++l;
    const uint64_t m = ((1ull << (31 + l)) + d - 1) / d;
    return MagicDiv{static_cast<uint32_t>(m), l - 1};
}

/// Everything the kernels need to know about one scan of one image: a launch covers one job (passed
/// by value, the drop-in API) or an array of jobs, one per blockIdx.y (the batch API).

struct ScanJob {
    const uint8_t* bytes;        // transferred entropy-coded bytes of the image
    const DestuffChunk* chunks;
    const Segment* segments;
    const uint8_t* tables;       // Huffman table pack of the scan (write pass)
    const uint8_t* tables_sync;  // sync pack of the scan (state-only passes)
    const uint16_t* qtables;     // uint16[4][64], natural order
    uint8_t* destuffed;
    int* seg_idx;                // subsequence -> segment
    int* st_p;                   // sync state, structure of arrays (reference `subsequence_info`,
    int* st_n;                   //   src/decode_huffman.cu:71-89, plus the DC sums)
    int* st_cz;                  // c | z << 8
    uint32_t* st_dc01;           // wrapping 16-bit sums of committed DC differences, components 0 | 1 << 16
    uint32_t* st_dc23;           // components 2 | 3 << 16
    uint8_t* pending;            // [num_subseq] 1: a flow that left subsequence i is unfinished (huff_sync_tail)
    int* flow_list;              // [num_subseq] scratch of huff_sync_tail
    const int* tail_parts;       // [num_tail_parts + 1] subsequence ranges of huff_sync_tail's workgroups
    int num_tail_parts;
    int max_tail_part;           // subsequences in the largest part
    uint32_t* fuse_ctl;          // control words of huff_tail_write (jg_kernels.hip: the tail kernel and the write pass in ONE launch), set
                                 //   up by the sequence kernel of the same call: kFuseCtlWords(sequences) of them
    int* tails_n;                // per-sequence aggregates used to place the write pass
    uint32_t* tails_dc01;
    uint32_t* tails_dc23;
    int* mh_p;                   // multi-hypothesis speculation (above): candidate states [mh][num_subseq],
    int* mh_cz;
    uint32_t* mh_link;           //   their links,
    uint2_t* mh_pool;            //   [1 + mh_pool_entries]: entry 0.x counts the states handed out, states from entry 1 on
    uint8_t* mh_known;           //   [num_subseq]: the table entry is a state of the resolved chain
    const MhBlock* mh_blocks;    //   block-wise chain walk (segments longer than kMhMaxSegSubseq): [num_mh_blocks], else null
    uint16_t* mh_blk_exit;       //   [num_mh_blocks][64]: where the chain leaves a block it enters at node h << 3 | row
    uint16_t* mh_blk_entry;      //   [num_mh_blocks]: the node it really enters at, or kMhBlockBroken
    int num_mh_blocks;
    uint16_t* sym;               // symbol stream: one region of `sym_region` 16-bit entries per subsequence, interleaved (above)
    uint2_t* du_tab;             // per data unit (stream order): {physical index of the first entry, number of entries}
    uint32_t sym_region;         // entries per subsequence region
    uint64_t sym_entries;        // entries of the buffer (sym_buffer_entries)
    int num_chunks;
    int num_seq;
    ScanParams sp;
    IdctParams ip;
    int* bnd_p;                  // [num_seq] exit state of the subsequence in front of sequence b as b's workgroup assumed it
    int* bnd_cz;                 //   (-1: nothing to say): a boundary where it equals the stored state needs no flow
};

} // namespace jg

#endif // JG_DEFS_H_
