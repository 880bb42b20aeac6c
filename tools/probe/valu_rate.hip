// Issue rate of integer VALU instructions on gfx950 against waves per SIMD (tools/probe, not part of the product).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 11, a5 = a0 ^ 13, a6 = a0 + 17, a7 = a0 + 19;
    uint64_t w0 = a0, w1 = a1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) { // v_add_u32, independent chains
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(a5));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(a6));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(a7));
            } else if (KIND == 1) { // logic / shifts
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(a0) : "v"(a4));
                asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a1));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a2) : "v"(a6));
                asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(a3));
            } else if (KIND == 2) { // 64-bit shifts
                asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w0));
                asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w1));
                asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w0));
                asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w1));
            } else if (KIND == 3) { // dependent chain of adds
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a4));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a5));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a6));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a7));
            } else if (KIND == 4) { // cndmask + cmp
                asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(a4) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(a5) : "vcc");
                asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a2), "v"(a6) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(a7) : "vcc");
            } else if (KIND == 5) { // mul_lo_u32 / mul_u32_u24
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(a4));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(a5));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(a6));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(a7));
            } else if (KIND == 6) { // three-operand ops
                asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a0) : "v"(a4), "v"(a5));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a1) : "v"(a5));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(a6), "v"(a7));
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(a7), "v"(a4));
            } else { // packed 16-bit
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a0) : "v"(a4));
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a1) : "v"(a5));
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a2) : "v"(a6));
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a3) : "v"(a7));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ static_cast<uint32_t>(w0) ^ static_cast<uint32_t>(w1);
}

template <int KIND>
void run(const char* name, uint32_t* d_out, int cus)
{
    const int iters = 2000;
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        const int grid = cus * wgs_per_cu;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        spin<KIND><<<grid, 256>>>(d_out, 10, 1);
        CHECK(hipEventRecord(e0));
        spin<KIND><<<grid, 256>>>(d_out, iters, 1);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double insts_per_wave = double(iters) * 16 * 4;
        const double wave_insts     = insts_per_wave * grid * 4;
        // per SIMD: wgs_per_cu waves; cycles at 2.4 GHz nominal
        const double cyc_per_inst_per_simd = ms * 1e-3 * 2.4e9 / (insts_per_wave * wgs_per_cu);
        std::printf("%-14s waves/SIMD %d: %.3f ms, %.1f G wave-instr/s, %.2f cycles (at 2.4 GHz) per wave-instruction per SIMD\n", name,
                    wgs_per_cu, ms, wave_insts / ms / 1e6, cyc_per_inst_per_simd);
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("%s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, size_t(cus) * 8 * 256 * 4));
    run<0>("v_add_u32", d_out, cus);
    run<1>("logic/shift", d_out, cus);
    run<2>("shift_b64", d_out, cus);
    run<3>("dependent add", d_out, cus);
    run<4>("cmp+cndmask", d_out, cus);
    run<5>("mul lo/u24", d_out, cus);
    run<6>("3-operand", d_out, cus);
    run<7>("v_pk_add_u16", d_out, cus);
    return 0;
}
