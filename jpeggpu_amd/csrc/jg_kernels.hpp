// jg_kernels.hpp -- launch interface of the gfx950 kernels (all launches are asynchronous on `stream`).
#ifndef JG_KERNELS_HPP_
#define JG_KERNELS_HPP_

#include "jg_defs.h"

#include <hip/hip_runtime_api.h>

namespace jg {

bool subseq_bytes_supported(int subseq_bytes);

/// Stages of one decode, in launch order (also the indices of jpeggpu_ext_get_stage_ms).
enum Stage {
    kStageFront     = 0, // device-side marker scan (jg_front.hip), launched by the decoder; nothing to launch here
    kStageDestuff   = 1,
    kStageSyncIntra = 2,
    kStageSyncInter = 3, // huff_sync_tail: sequence boundaries + flows the intra kernel left unfinished
    kStageTails     = 4,
    kStageWrite     = 5,
    kStageIdct      = 6,
    kNumStages      = 7
};

/// Work extents of a launch: the largest count over the jobs it covers (blocks beyond a job's own
/// count exit at once).
struct JobExtent {
    int max_chunks      = 0;
    int max_seq         = 0;
    int max_tail_parts  = 0;
    int max_tail_part   = 0; // subsequences in the largest tail part of any job
    int max_idct_blocks = 0;
    int subseq_words    = 0; // identical for every job of a launch
    uint32_t max_tab_bytes = 0;      // largest write-pass table pack
    uint32_t max_tab_bytes_sync = 0; // largest sync pack
    bool fuse_tail_write = false;    // batch launches: the tail kernel's parts and the write pass's sequences as ONE launch
                                     // (huff_tail_write), launched at kStageWrite; kStageSyncInter and kStageTails launch nothing
    bool repack_flows   = false;     // batch launches: the lone decode's sequence kernel (every flow kept in its workgroup), for
                                     // calls too small to fill the chip; the jobs then carry max_intra_iters = kSeqLanes, tail_marks = 0
};
void extend(JobExtent& e, const ScanJob& job);
/// Writers of huff_tail_write that gave up waiting (0 on a correct run), since the library was loaded.
hipError_t read_fuse_timeouts(unsigned int* count);

/// One stage for ONE job passed by value as a kernel argument (the drop-in single-image API).
hipError_t launch_stage(Stage stage, const ScanJob& job, hipStream_t stream);

/// One stage for ALL scans of one image (1..kMaxScans jobs, by value, one per blockIdx.y; the same subsequence size).
hipError_t launch_stage_scans(Stage stage, const ScanJob* jobs, int num_jobs, hipStream_t stream);

/// The three kernels of the multi-hypothesis speculation (jg_defs.h) for ONE job with sp.mh > 1, in front of
/// kStageSyncIntra. `max_seg_subseq`: subsequences of the job's largest restart segment (kMhMaxSegSubseq for a
/// device-scanned image, whose segments the host does not know: longer ones fall back on the device). `d_job`: the
/// job's copy in device memory if the device-side front end has filled in its counts, else null.
hipError_t launch_mh(const ScanJob& job, const ScanJob* d_job, int max_seg_subseq, hipStream_t stream);

/// One stage for ONE job stored in device memory (the lone decode of a device-scanned image: the device-side front end
/// has filled in its counts; `extent` holds the header's upper bounds), with the kernels' lone-decode variants.
hipError_t launch_stage_device_job(Stage stage, const ScanJob* d_job, const JobExtent& extent, hipStream_t stream);

/// One stage for `num_jobs` jobs stored in device memory, one per blockIdx.y (the batch API).
hipError_t launch_stage_batch(
    Stage stage, const ScanJob* d_jobs, int num_jobs, const JobExtent& extent, hipStream_t stream);

/// Nearest-neighbour replication of one plane: dst[y][x] = src[y * num_y / den_y][x * num_x / den_x]
/// (integer part of the reference's host helper util/util.h:62-91).
hipError_t launch_upsample(
    const uint8_t* src, int src_pitch, int src_w, int src_h,
    uint8_t* dst, int dst_pitch, int dst_w, int dst_h,
    int num_x, int den_x, int num_y, int den_y, hipStream_t stream);

/// Nearest-neighbour replication + YCbCr -> interleaved RGB8 (reference host helper util/util.h:62-104);
/// `ncomp` 1 (grey copied to R, G, B) or 3.
hipError_t launch_rgbi(
    const uint8_t* const* planes, const int* pitch, const int* w, const int* h, const int* num_x, const int* num_y,
    int den_x, int den_y, int ncomp, uint8_t* dst, int dst_pitch, int width, int height, hipStream_t stream);

} // namespace jg

#endif // JG_KERNELS_HPP_
