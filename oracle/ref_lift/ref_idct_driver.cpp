// ref_idct_driver.cpp -- C entry points around the reference's own IDCT functions.
//
// TEST INFRASTRUCTURE. "idct_lifted.inc" is /root/reference/src/idct.cu:43-144, extracted by build.sh at build
// time into a temporary directory (never committed, never shipped): unfixh, unfixo, idct_vector, idct_col, idct_row, compiled as they
// stand with -D__device__= . Everything arithmetic below is THEIR code; this file only moves data.
//
// What cannot be lifted is the body of `idct_kernel` (idct.cu:146-223, a __global__ function indexed by
// threadIdx): its three arithmetic statements are restated in ref_idct_block() next to the lines they follow.
#include <stdint.h>

#include <algorithm>

namespace {
#include "idct_lifted.inc"
}

extern "C" {

/// idct_vector on n 8-vectors: in/out int32[n][8] (idct.cu:50-95; results are unfixh'd int16 values).
void ref_idct_vectors(const int32_t* in, int32_t* out, int n)
{
    for (int i = 0; i < n; ++i) {
        int v[8];
        for (int k = 0; k < 8; ++k) v[k] = in[8 * i + k];
        idct_vector(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        for (int k = 0; k < 8; ++k) out[8 * i + k] = v[k];
    }
}

/// One data unit the way idct_kernel handles it. coef: quantised coefficients, natural (row-major) order;
/// q: quantisation table, natural order; signed_q != 0: the reference's literal `const int8_t qval`
/// (idct.cu:179), 0: the value as the unsigned byte T.81 defines (this repository's
/// documented deviation; identical for q <= 127).
void ref_idct_block(const int16_t* coef, const uint8_t* q, uint8_t* out, int signed_q)
{
    // idct.cu:160-161: int16_t block[shared_stride * ...], one 8x8 unit of it here; rows are 4-byte aligned
    // there as well (idct_row reads them as uint32_t)
    alignas(4) int16_t block[64];
    for (int i = 0; i < 8; ++i)      // idct.cu:173-181
        for (int x = 0; x < 8; ++x) {
            const int16_t val = coef[i * 8 + x];
            if (signed_q) {
                const int8_t qval = static_cast<int8_t>(q[i * 8 + x]); // idct.cu:179
                block[i * 8 + x]  = val * qval;                          // idct.cu:180 (int16 store)
            } else {
                const uint8_t qval = q[i * 8 + x];
                block[i * 8 + x]   = val * qval;
            }
        }
    for (int x = 0; x < 8; ++x) idct_col(block + x, 8);                                       // idct.cu:185-187
    for (int y = 0; y < 8; ++y) idct_row(reinterpret_cast<uint32_t*>(block + y * 8));         // idct.cu:190-194
    for (int i = 0; i < 8; ++i)      // idct.cu:209-221
        for (int x = 0; x < 8; ++x) {
            const int16_t val = block[i * 8 + x] + 128;                                       // idct.cu:218
            out[i * 8 + x]    = static_cast<uint8_t>(std::max(0, std::min<int>(val, 255)));   // idct.cu:220
        }
}

void ref_idct_blocks(const int16_t* coef, const uint8_t* q, uint8_t* out, int n, int signed_q)
{
    for (int i = 0; i < n; ++i) ref_idct_block(coef + 64 * i, q, out + 64 * i, signed_q);
}

} // extern "C"
