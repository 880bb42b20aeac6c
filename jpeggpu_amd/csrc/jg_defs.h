// jg_defs.h -- plain-old-data shared by the host parser, the launch code and the gfx950 kernels.
//
// Vocabulary follows the reference (src/decoder.cpp:46-63): scan, segment (bytes between restart
// markers), subsequence (fixed slice of destuffed segment data decoded by one lane), data unit
// (8x8 block), MCU. A "sequence" here is the run of subsequences owned by one workgroup.
#ifndef JG_DEFS_H_
#define JG_DEFS_H_

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
constexpr int kHuffSlots    = 8;  // slot = Th*2 + Tc (Tc: 0 = DC, 1 = AC), reference src/reader.cpp:263
constexpr int kSeqSubseq    = 256; // subsequences per workgroup ("sequence"), reference decode_huffman.cu:777
constexpr int kDestuffWin   = 4096; // stuffed bytes handled by one destuff workgroup (256 lanes x 16 B)

/// Zig-zag index -> raster index inside a data unit (T.81 figure A.6; reference src/defs.hpp:94-102).
#define JG_ORDER_NATURAL                                                                           \
    {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,       \
     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,       \
     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63}

/// Device Huffman table. Same information as the reference's `huffman_table`
/// (src/reader.hpp:45-64: 8-bit LUT + maxcode/valptr walk + huffval) but the LUT entry is
/// pre-digested so the per-symbol step needs one LDS read and no table-class branch:
///   bits  0..4  code length 1..8, 0 = code longer than 8 bits (slow path)
///   bits  5..10 total symbol length = code length + magnitude bits
///   bits 11..14 magnitude category s (0..15)
///   bit  15     end-of-block (AC symbol with s == 0 and run != 15), always 0 in a DC table
///   bits 16..20 run + 1 (1..16); 1 in a DC table
struct HuffTableDev {
    uint32_t lut[256];
    int32_t maxcode[16]; // largest code of length l+1, -1 if none (reference reader.cpp:213-223)
    int32_t valoff[16];  // huffval index of first code of length l+1 minus that code
    uint8_t huffval[256];
};
static_assert(sizeof(HuffTableDev) == 1408, "layout is shared with the kernels");

JG_HD inline uint32_t huff_entry(int codelen_for_lut, int codelen, uint32_t sym, bool is_dc)
{
    // DC: sym is the category (hardened to 4 bits; valid baseline streams use 0..11).
    // AC: sym = run << 4 | category (reference decode_huffman.cu:232-259).
    const uint32_t s   = sym & 15u;
    const uint32_t r   = is_dc ? 0u : (sym >> 4);
    const uint32_t eob = (!is_dc && s == 0 && r != 15) ? 1u : 0u;
    return static_cast<uint32_t>(codelen_for_lut) | ((codelen + s) << 5) | (s << 11) | (eob << 15) |
           ((r + 1u) << 16);
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
    uint32_t du_comp;     // 2 bits per data unit of the MCU: scan-component index
    uint32_t dc_slot;     // 4 bits per scan component: Huffman slot of its DC table
    uint32_t ac_slot;     // 4 bits per scan component: Huffman slot of its AC table
};

/// Geometry for dequant + IDCT reading the stream-order coefficient buffer (replaces the reference's
/// separate transpose pass, src/decode_transpose.cu:41-132, plus src/idct.cu:146-223).
struct IdctParams {
    int num_du;     // data units in the scan
    int du_per_mcu;
    int mcus_x;
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

} // namespace jg

#endif // JG_DEFS_H_
