// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes the decode kernels use (the guide says the
// counter is exact only for what it was calibrated on: /opt/skills/guides/MI355X_MICROARCH.md, "HBM"). Every kernel
// below moves a KNOWN number of bytes of a buffer far larger than L2 + Infinity Cache exactly once; the ratio
// counter / known bytes is the factor tools/summarize_profiles.py applies to the kernel with that shape.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe/fetch_probe.hip -o gpurun_out/fetch_probe
//   cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_probe_rd -o p --output-format csv -- $OUT/fetch_probe
//   cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/fetch_probe_wr -o p --output-format csv -- $OUT/fetch_probe
//   python tools/probe/fetch_probe_summary.py $OUT/fetch_probe_rd $OUT/fetch_probe_wr   -> profiles/rNN_fetch_calibration.json
//
//   rd_wide16      16 B per lane, consecutive lanes consecutive (destuff_kernel's window loads)
//   rd_dword_rows  4 B per lane, lanes 0..31 one 128-byte line, lanes 32..63 a line one "tile" further, every step the
//                  next line of both (the bitstream refills of the Huffman kernels over the tiled rows)
//   rd_u16_units   2 B per lane, groups of 8 lanes read 16 consecutive bytes at the start of consecutive 32-byte
//                  sectors (the IDCT's gather of a unit's first entries): HALF of every sector is asked for
//   rd_rec8        8 B per lane, consecutive (data-unit records read by the IDCT)
//   wr_sector32    two 16-byte stores per lane into one 32-byte sector, sectors of the 64 lanes consecutive (the write
//                  pass's ring flushes and the unit records)
//   wr_row8        8 B per lane, lanes 0..31 consecutive, lanes 32..63 one pitch further (the IDCT's pixel rows)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void rd_wide16(const uint4* __restrict__ src, size_t n16, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// One wave walks `steps` lines of two tiles: bytes read per wave = steps * 256, every byte of the buffer once.
__global__ __launch_bounds__(256) void rd_dword_rows(const uint32_t* __restrict__ src, size_t n_bytes, int steps, uint32_t* sink)
{
    const size_t wave   = (blockIdx.x * 256ull + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t tile   = static_cast<size_t>(steps) * 128;         // bytes of one "tile": `steps` lines
    const size_t base   = wave * 2 * tile + (lane >> 5) * tile + (lane & 31u) * 4;
    if ((wave + 1) * 2 * tile > n_bytes) return;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) acc ^= src[(base + static_cast<size_t>(s) * 128) >> 2];
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void rd_u16_units(const uint16_t* __restrict__ src, size_t n_sectors, uint32_t* sink)
{
    uint32_t acc = 0;
    // group g of 8 lanes reads entries 0..7 of sector g
    for (size_t g = (blockIdx.x * 256ull + threadIdx.x) >> 3; g < n_sectors; g += gridDim.x * 32ull)
        acc ^= src[g * 16 + (threadIdx.x & 7u)];
    if (acc == 0x1234u) *sink = acc; // a 16-bit value: a 32-bit constant here lets the compiler drop the loop
}

__global__ __launch_bounds__(256) void rd_rec8(const uint2* __restrict__ src, size_t n8, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n8; i += gridDim.x * 256ull) {
        const uint2 v = src[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void wr_sector32(uint4* __restrict__ dst, size_t n_sectors)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_sectors; i += gridDim.x * 256ull) {
        const uint32_t x = static_cast<uint32_t>(i);
        dst[2 * i]     = make_uint4(x, x + 1, x + 2, x + 3);
        dst[2 * i + 1] = make_uint4(x + 4, x + 5, x + 6, x + 7);
    }
}

__global__ __launch_bounds__(256) void wr_row8(uint2* __restrict__ dst, size_t n_rows /* pairs of 256-byte half rows */, size_t pitch8)
{
    // wave w stores 8 bytes per lane: lanes 0..31 at row 2w, lanes 32..63 at row 2w + 1 (rows `pitch8` uint2 apart)
    const size_t wave   = (blockIdx.x * 256ull + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= n_rows) return;
    const size_t col_block = wave % (pitch8 / 32), row_pair = wave / (pitch8 / 32);
    dst[(2 * row_pair + (lane >> 5)) * pitch8 + col_block * 32 + (lane & 31u)] = make_uint2(lane, static_cast<uint32_t>(wave));
}

int main()
{
    const size_t bytes = 2ull << 30; // 2 GiB: eight times the Infinity Cache
    void* buf          = nullptr;
    uint32_t* sink     = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        rd_wide16<<<grid, 256>>>(static_cast<const uint4*>(buf), bytes / 16, sink);
        const int steps = 64;
        const size_t waves = bytes / (2ull * steps * 128);
        rd_dword_rows<<<static_cast<unsigned>((waves + 3) / 4), 256>>>(static_cast<const uint32_t*>(buf), bytes, steps, sink);
        rd_u16_units<<<grid, 256>>>(static_cast<const uint16_t*>(buf), bytes / 32, sink);
        rd_rec8<<<grid, 256>>>(static_cast<const uint2*>(buf), bytes / 8, sink);
        wr_sector32<<<grid, 256>>>(static_cast<uint4*>(buf), bytes / 32);
        const size_t pitch8 = 4096 / 8, rows = bytes / 4096 / 2;
        wr_row8<<<static_cast<unsigned>((rows * (pitch8 / 32) + 3) / 4), 256>>>(static_cast<uint2*>(buf), rows * (pitch8 / 32), pitch8);
        CK(hipDeviceSynchronize());
    }
    std::printf("{\"buffer_bytes\": %zu, \"known_bytes\": {\"rd_wide16\": %zu, \"rd_dword_rows\": %zu, \"rd_u16_units_asked\": %zu, "
                "\"rd_u16_units_sectors\": %zu, \"rd_rec8\": %zu, \"wr_sector32\": %zu, \"wr_row8\": %zu}}\n",
                bytes, bytes, bytes, bytes / 2, bytes, bytes, bytes, bytes);
    CK(hipFree(buf));
    CK(hipFree(sink));
    return 0;
}
