// Per-instruction issue cost of integer VALU instructions on gfx950, 4 waves per SIMD, independent streams
// (tools/probe, not part of the product).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

#define OPS(X) \
    X(0, "v_add_u32 %0, %0, %2") \
    X(1, "v_sub_u32 %0, %0, %2") \
    X(2, "v_and_b32 %0, %0, %2") \
    X(3, "v_or_b32 %0, %0, %2") \
    X(4, "v_xor_b32 %0, %0, %2") \
    X(5, "v_lshlrev_b32 %0, 3, %0") \
    X(6, "v_lshrrev_b32 %0, 3, %0") \
    X(7, "v_lshlrev_b32 %0, %2, %0") \
    X(8, "v_ashrrev_i32 %0, 3, %0") \
    X(9, "v_bfe_u32 %0, %0, 3, 20") \
    X(10, "v_bfe_u32 %0, %0, %2, %2") \
    X(11, "v_min_u32 %0, %0, %2") \
    X(12, "v_max_i32 %0, %0, %2") \
    X(13, "v_mov_b32 %0, %2") \
    X(14, "v_not_b32 %0, %0") \
    X(15, "v_bcnt_u32_b32 %0, %0, %2") \
    X(16, "v_mul_u32_u24 %0, %0, %2") \
    X(17, "v_mul_i32_i24 %0, %0, %2") \
    X(18, "v_mul_lo_u32 %0, %0, %2") \
    X(19, "v_mad_u32_u24 %0, %0, %2, %2") \
    X(20, "v_add3_u32 %0, %0, %2, %2") \
    X(21, "v_lshl_add_u32 %0, %0, 2, %2") \
    X(22, "v_add_lshl_u32 %0, %0, %2, 2") \
    X(23, "v_lshl_or_b32 %0, %0, 2, %2") \
    X(24, "v_and_or_b32 %0, %0, %2, %2") \
    X(25, "v_or3_b32 %0, %0, %2, %2") \
    X(26, "v_xad_u32 %0, %0, %2, %2") \
    X(27, "v_bfi_b32 %0, %0, %2, %2") \
    X(28, "v_perm_b32 %0, %0, %2, %2") \
    X(29, "v_alignbit_b32 %0, %0, %2, 8") \
    X(30, "v_alignbyte_b32 %0, %0, %2, 1") \
    X(31, "v_pk_add_u16 %0, %0, %2") \
    X(32, "v_pk_mul_lo_u16 %0, %0, %2") \
    X(33, "v_pk_lshlrev_b16 %0, 1, %0") \
    X(34, "v_pk_max_i16 %0, %0, %2") \
    X(35, "v_sat_pk_u8_i16 %0, %0") \
    X(36, "v_cndmask_b32 %0, %0, %2, vcc") \
    X(37, "v_cmp_lt_u32 vcc, %0, %2") \
    X(38, "v_add_co_u32 %0, vcc, %0, %2") \
    X(39, "v_ffbh_u32 %0, %0") \
    X(40, "v_add_u16 %0, %0, %2") \
    X(41, "v_add_f32 %0, %0, %2") \
    X(42, "v_fma_f32 %0, %0, %2, %2") \
    X(43, "v_med3_i32 %0, %0, %2, %2") \
    X(44, "v_sub_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD") \
    X(45, "v_add_u32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(46, "v_mad_u64_u32 %1, vcc, %0, %2, %1") \
    X(47, "v_lshlrev_b64 %1, 3, %1") \
    X(48, "v_lshrrev_b64 %1, %2, %1") \
    X(49, "v_lshl_add_u64 %1, %1, 2, %1") \
    X(50, "v_bitop3_b32 %0, %0, %2, %2 bitop3:0x96") \
    X(51, "v_cvt_f32_u32 %0, %0") \
    X(52, "v_mul_hi_u32 %0, %0, %2") \
    X(53, "v_readfirstlane_b32 s20, %0") \
    X(54, "v_mad_i32_i24 %0, %0, %2, %2") \
    X(55, "v_mul_i32_i24 %0, 0x5a82, %0") \
    X(56, "v_mul_i32_i24_sdwa %0, sext(%0), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD") \
    X(57, "v_add_u32_sdwa %0, sext(%0), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0") \
    X(58, "v_add_u32 %0, 0x8000, %0") \
    X(59, "v_bfe_i32 %0, %0, 0, 16") \
    X(60, "v_and_b32 %0, -4, %0")

template <int OP>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters)
{
    uint32_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = a0 ^ 11;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#define X(id, text)                                                                    \
    if (OP == id) {                                                                    \
        asm volatile(text : "+v"(a0), "+v"(w0) : "v"(b) : "vcc", "s20");               \
        asm volatile(text : "+v"(a1), "+v"(w1) : "v"(b) : "vcc", "s20");               \
        asm volatile(text : "+v"(a2), "+v"(w2) : "v"(b) : "vcc", "s20");               \
        asm volatile(text : "+v"(a3), "+v"(w3) : "v"(b) : "vcc", "s20");               \
    }
            OPS(X)
#undef X
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ static_cast<uint32_t>(w0 ^ w1 ^ w2 ^ w3);
}

template <int OP>
void run(const char* name, uint32_t* d_out, int cus, double base_ms)
{
    const int iters = 4000, wgs_per_cu = 4, grid = cus * wgs_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    spin<OP><<<grid, 256>>>(d_out, 10);
    CHECK(hipEventRecord(e0));
    spin<OP><<<grid, 256>>>(d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_wave = double(iters) * 16 * 4;
    std::printf("%-100s %.3f ms  %.2f cycles/wave-instr/SIMD at 2.4 GHz  (%.2f x v_add_u32)\n", name, ms,
                ms * 1e-3 * 2.4e9 / (insts_per_wave * wgs_per_cu), base_ms > 0 ? ms / base_ms : 1.0);
    if (OP == 0) *reinterpret_cast<double*>(&base_ms) = ms;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, size_t(cus) * 8 * 256 * 4));
    double base = 0;
    {
        // measure the base once
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        spin<0><<<cus * 4, 256>>>(d_out, 10);
        CHECK(hipEventRecord(e0));
        spin<0><<<cus * 4, 256>>>(d_out, 4000);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        base = ms;
    }
#define X(id, text) run<id>(text, d_out, cus, base);
    OPS(X)
#undef X
    return 0;
}
