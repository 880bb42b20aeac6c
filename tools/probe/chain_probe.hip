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
    } else if (mode == 3) {
        // four VALU-compare -> scalar-branch round trips per iteration (never taken), 4 VALU besides
        for (int i = 0; i < n_iter; ++i) {
            asm volatile(
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_eq_u32 vcc, 0xdead, %0\n\t"
                "s_cbranch_vccnz 7f\n\t"
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_eq_u32 vcc, 0xdead, %0\n\t"
                "s_cbranch_vccnz 7f\n\t"
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_eq_u32 vcc, 0xdead, %0\n\t"
                "s_cbranch_vccnz 7f\n\t"
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_eq_u32 vcc, 0xdead, %0\n\t"
                "s_cbranch_vccnz 7f\n"
                "7:\n\t"
                : "+v"(x) : : "vcc");
        }
    } else if (mode == 4) {
        // the same four compares feeding exec-mask updates (s_and_b64 exec) and execz branches
        for (int i = 0; i < n_iter; ++i) {
            asm volatile(
                "s_mov_b64 s[20:21], exec\n\t"
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_ne_u32 vcc, 0xdead, %0\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 8f\n\t"
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_ne_u32 vcc, 0xdead, %0\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 8f\n\t"
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_ne_u32 vcc, 0xdead, %0\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 8f\n\t"
                "v_add_u32 %0, 1, %0\n\t"
                "v_cmp_ne_u32 vcc, 0xdead, %0\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 8f\n"
                "8:\n\t"
                "s_mov_b64 exec, s[20:21]\n\t"
                : "+v"(x) : : "vcc", "s20", "s21");
        }
    } else if (mode == 6) {
        // the step of the hand-scheduled symbol loop (jg_huff_core.h, sync_steps_asm), every look-up a plain hit
        uint32_t hi = x * 2654435761u, lo = x * 40503u + 7u, sh = 5, zm = 3, p = 0, isdc = 0, shift = 21, tab = 0, plast = 0x7fffffff;
        uint32_t peek, e, t, zp, total, u, zold;
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0x0a050a05u; // length 5, advance 5 (hi = lo half)
        __syncthreads();
        for (int i = 0; i < n_iter; ++i) {
            asm volatile(
                "v_alignbit_b32 %[peek], %[hi], %[lo], %[sh]\n\t"
                "v_lshrrev_b32 %[u], %[shift], %[peek]\n\t"
                "v_lshl_add_u32 %[u], %[u], 2, %[tab]\n\t"
                "ds_read_b32 %[e], %[u]\n\t"
                "v_mov_b32 %[zold], %[zm]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_bfe_u32 %[t], %[e], 21, 4\n\t"
                "v_add_u32 %[t], %[zm], %[t]\n\t"
                "v_cmp_gt_i32 vcc, 63, %[t]\n\t"
                "v_and_b32 %[t], 31, %[e]\n\t"
                "v_add_u32 %[t], -1, %[t]\n\t"
                "v_cndmask_b32_sdwa %[e], %[e], %[e], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                "v_lshrrev_b32 %[zp], 9, %[e]\n\t"
                "v_add_u32 %[zm], %[zm], %[zp]\n\t"
                "v_and_b32 %[zm], 31, %[zm]\n\t"
                "v_and_b32 %[total], 31, %[e]\n\t"
                "v_sub_u32 %[sh], %[sh], %[total]\n\t"
                "v_and_b32 %[sh], 31, %[sh]\n\t"
                "v_add_u32 %[p], %[p], %[total]\n\t"
                "v_sub_u32 %[zp], 62, %[zm]\n\t"
                "v_or3_b32 %[t], %[t], %[zp], %[isdc]\n\t"
                "v_sub_u32 %[zp], %[plast], %[p]\n\t"
                "v_or_b32 %[t], %[t], %[zp]\n\t"
                "v_or_b32 %[zp], %[t], %[sh]\n\t"
                "v_cmp_gt_i32 vcc, 0, %[zp]\n\t"
                "s_cbranch_vccnz 7f\n"
                "7:\n\t"
                : [hi] "+v"(hi), [lo] "+v"(lo), [sh] "+v"(sh), [zm] "+v"(zm), [p] "+v"(p), [peek] "=&v"(peek), [e] "=&v"(e), [t] "=&v"(t),
                  [zp] "=&v"(zp), [total] "=&v"(total), [u] "=&v"(u), [zold] "=&v"(zold)
                : [isdc] "v"(isdc), [shift] "v"(shift), [tab] "v"(tab), [plast] "v"(plast)
                : "vcc");
        }
        x = p + zm + sh + zold;
    } else if (mode == 5) {
        // eight independent-looking VALU adds per iteration, for the plain issue rate
        for (int i = 0; i < n_iter; ++i) {
            asm volatile("v_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\t"
                         "v_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\t" : "+v"(x));
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
    for (int blocks : {206}) {
        for (int mode : {0, 3, 5, 6}) {
            const int n = 20000;
            for (int rep = 0; rep < 2; ++rep) {
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
