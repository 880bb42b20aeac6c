// jg_front.hpp -- launch interface of the device-side front end (jg_front.hip).
#ifndef JG_FRONT_HPP_
#define JG_FRONT_HPP_

#include "jg_defs.h"

#include <hip/hip_runtime_api.h>

namespace jg {

/// Everything the front-end kernels need; all pointers are device memory inside d_tmp.
struct FrontParams {
    const uint8_t* bytes;      // transferred bytes; offset 0 is the origin of the 4 KiB window grid
    uint32_t bytes_len;        // valid bytes
    uint32_t scan_begin;       // offset of the first entropy-coded byte
    uint32_t num_windows;
    uint32_t expect_segments;  // ceil(MCUs / restart interval), from the frame and DRI headers
    uint32_t subseq_bytes;
    uint32_t max_subseq;       // capacities computed from the header
    uint32_t max_chunks;
    uint32_t max_parts;        // entries of tail_parts (parts + 1)
    uint32_t* win_data;        // [num_windows]      data bytes per window
    uint32_t* win_nmark;       // [num_windows]      markers per window
    uint32_t* win_bad;         // [num_windows]      position of the first FF FF 00 of the window, or 0xFFFFFFFF
    uint32_t* win_prefix;      // [num_windows + 1]
    uint32_t* mark_off;        // [num_windows + 1]
    uint32_t* mk_pos;          // [expect_segments + 1]  position of the i-th marker of the scan
    uint32_t* mk_g;            // [expect_segments + 1]  data bytes of the scan in front of it
    uint32_t* seg_cnt;         // [expect_segments + 1]
    uint32_t* seg_nch;         // [expect_segments + 1]
    Segment* segments;         // [expect_segments]          out
    DestuffChunk* chunks;      // [max_chunks]               out
    int* tail_parts;           // [max_parts]                out
    MhBlock* mh_blocks;        // [max_mh_blocks] out, or null: blocks of the multi-hypothesis chain walk of a scan WITHOUT restart
    uint32_t max_mh_blocks;    //   markers (one segment: blocks of kMhMaxSegSubseq subsequences, jg_defs.h)
    ScanJob* job;              // the scan's job in device memory: counts are filled in
    uint32_t* status;          // [8]: jpeggpu_status, subsequences, segments, chunks, tail parts, -, first FF FF 00, terminator ordinal
};

/// One scan, parameters by value. `job` is stored to P.job by the first kernel (a kernel argument is captured at
/// launch, so the caller's copy may change as soon as this returns -- no staging buffer, no copy from pageable memory).
hipError_t launch_front(const FrontParams& P, const ScanJob& job, hipStream_t stream);
/// The same for `count` scans whose parameters sit in device memory (grid.y = scan).
hipError_t launch_front_batch(const FrontParams* d_params, int count, uint32_t max_windows, hipStream_t stream);

} // namespace jg

#endif // JG_FRONT_HPP_
