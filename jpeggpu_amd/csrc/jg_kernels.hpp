// jg_kernels.hpp -- launch interface of the gfx950 kernels (all launches are asynchronous on `stream`).
#ifndef JG_KERNELS_HPP_
#define JG_KERNELS_HPP_

#include "jg_defs.h"

#include <hip/hip_runtime_api.h>

namespace jg {

/// Per-subsequence synchronisation state in device memory, structure-of-arrays
/// (reference `subsequence_info`, src/decode_huffman.cu:71-89, plus the DC sums).
struct SubseqState {
    int* p;             // bit position after the last committed symbol, relative to the segment
    int* n;             // coefficient slots committed
    int* cz;            // c | z << 8
    uint32_t* dc01;     // wrapping 16-bit sums of committed DC differences, scan components 0 | 1 << 16
    uint32_t* dc23;     // scan components 2 | 3 << 16
};

/// Per-sequence (workgroup) aggregate used to place the write pass without a device-wide scan.
struct SeqTails {
    int* n;
    uint32_t* dc01;
    uint32_t* dc23;
};

bool subseq_bytes_supported(int subseq_bytes);

hipError_t launch_destuff(
    const uint8_t* d_bytes,
    uint8_t* d_destuffed,
    int* d_seg_idx,
    const DestuffChunk* d_chunks,
    int num_chunks,
    int subseq_bytes,
    hipStream_t stream);

/// The four Huffman launches of one scan, in order. `which` selects one of them so the caller can
/// place timing events between the kernels.
enum HuffStage { kHuffSyncIntra = 0, kHuffSyncInter = 1, kHuffTails = 2, kHuffWrite = 3 };

hipError_t launch_huffman_stage(
    HuffStage which,
    const uint8_t* d_destuffed,
    const Segment* d_segments,
    const int* d_seg_idx,
    const uint8_t* d_tables, // the scan's table pack (jg_defs.h)
    const ScanParams& sp,
    SubseqState st,
    SeqTails tails,
    int16_t* d_coef, // stream-order coefficients, must be zero-filled
    hipStream_t stream);

hipError_t launch_idct(
    const int16_t* d_coef, const uint8_t* d_qtables, const IdctParams& ip, hipStream_t stream);

/// Nearest-neighbour replication of one plane: dst[y][x] = src[y * num_y / den_y][x * num_x / den_x]
/// (integer part of the reference's host helper util/util.h:62-91).
hipError_t launch_upsample(
    const uint8_t* src, int src_pitch, int src_w, int src_h,
    uint8_t* dst, int dst_pitch, int dst_w, int dst_h,
    int num_x, int den_x, int num_y, int den_y, hipStream_t stream);

} // namespace jg

#endif // JG_KERNELS_HPP_
