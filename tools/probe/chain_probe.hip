// Latency of dependent chains on an otherwise idle chip: what one lane pays per dependent VALU op and per dependent
// LDS read when a single wave per SIMD runs (the single-image case), and the shader clock it runs at.
// hipcc --offload-arch=gfx950 -O3 tools/probe/chain_probe.hip -o gpurun_out/chain_probe && gpurun_out/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void probe(uint64_t* out, int n_iter, int mode)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (i * 2654435761u) & 4095u;
    __syncthreads();
    uint32_t x = threadIdx.x;
    const uint64_t c0 = __builtin_readcyclecounter(); // s_memtime
    const uint64_t r0 = wall_clock64();               // s_memrealtime, 100 MHz
    if (mode == 0) {
        for (int i = 0; i < n_iter; ++i) x = lds[x & 4095u]; // dependent LDS reads
    } else if (mode == 1) {
        for (int i = 0; i < n_iter; ++i) { // 8 dependent VALU ops per iteration
            x = x * 3u + 1u; x ^= x >> 3; x = x * 5u + 7u; x ^= x >> 5;
            asm volatile("" : "+v"(x));
        }
    } else {
        for (int i = 0; i < n_iter; ++i) { // LDS read + 8 VALU + a divergent-looking branch
            x = lds[x & 4095u];
            x = x * 3u + 1u; x ^= x >> 3;
            if (x & 1u) x = x * 5u + 7u;
            x ^= x >> 5;
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    const uint64_t r1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x + 0] = c1 - c0;
        out[3 * blockIdx.x + 1] = r1 - r0;
        out[3 * blockIdx.x + 2] = x;
    }
    if (x == 0xdeadbeef) out[0] = x;
}

int main()
{
    uint64_t* d;
    const int max_blocks = 4096;
    hipMalloc(&d, max_blocks * 3 * sizeof(uint64_t));
    std::vector<uint64_t> h(max_blocks * 3);
    for (int blocks : {1, 206, 2048}) {
        for (int mode = 0; mode < 3; ++mode) {
            const int n = 20000;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                probe<<<blocks, 256>>>(d, n, mode);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h.data(), d, blocks * 3 * sizeof(uint64_t), hipMemcpyDeviceToHost);
                printf("blocks %4d mode %d rep %d: kernel %.1f us, block0 memtime %llu ticks, realtime %llu ticks (%.1f us), per iter %.1f ns\n",
                       blocks, mode, rep, ms * 1e3, (unsigned long long)h[0], (unsigned long long)h[1], h[1] / 100.0, h[1] * 10.0 / n);
            }
        }
    }
    return 0;
}
