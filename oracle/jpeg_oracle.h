/* jpeg_oracle.h -- interface of the CPU oracle (test infrastructure, see jpeg_oracle.c). */
#ifndef JPEG_ORACLE_H_
#define JPEG_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as enum jpeggpu_status (reference include/jpeggpu/jpeggpu.h:38-52) */
enum {
    JO_OK               = 0,
    JO_INVALID_ARGUMENT = 1,
    JO_INVALID_JPEG     = 2,
    JO_INTERNAL         = 3,
    JO_NOT_SUPPORTED    = 4,
    JO_NO_MEMORY        = 5,
    JO_INCOMPLETE       = 6
};

/* reproduce the reference's signed read of quantiser values (src/idct.cu:179) */
#define JO_QUIRK_SIGNED_Q 1

typedef struct {
    int width, height, ncomp, restart_interval, nscans;
    int hs[4], vs[4], qidx[4];
    int plane_w[4], plane_h[4];   /* ceil(W*h/hmax) x ceil(H*v/vmax) */
    int blocks_w[4], blocks_h[4]; /* coefficient array size in blocks (rounded to the scan's MCU) */
    int16_t* coef[4];             /* [blocks_h][blocks_w][64], natural order, quantised, DC absolute */
    uint8_t* plane[4];            /* [plane_h][plane_w] */
    uint16_t qtab[4][64];         /* natural order; 8- or 16-bit entries in the file */
    /* per scan: coefficients in stream order (data unit after data unit as coded) */
    int16_t* stream_coef[4];
    int stream_du[4];
    int scan_ncomp[4];
    int scan_du_per_mcu[4];
    int scan_comp[4][4];
} jo_image;

int jo_decode(const uint8_t* data, size_t size, jo_image* img, int flags);
void jo_free(jo_image* img);

/* dequant + IDCT + level shift + clamp of one data unit (natural order in, raster out) */
void jo_idct_block(const int16_t coef[64], const uint16_t q[64], uint8_t out[64], int flags);

/* the 8-point fixed-point transform (reference idct_vector, src/idct.cu:50-95) on n vectors of 8 */
void jo_idct_vectors(const int32_t* in, int32_t* out, int n);

/* one Huffman symbol per 32-bit window / the byte rule of the destuffing, for the reference-built known answers */
int jo_symbol_steps(const uint8_t bits[16], const uint8_t* vals, int count, int is_dc, const uint32_t* win, const int* z, int n,
                    int* length, int* symbol, int* run);
void jo_byte_rule(const uint8_t* prev, const uint8_t* byte, int n, uint8_t* is_data, uint8_t* written);

typedef struct {
    int num_subseq, num_segments, num_du;
    size_t scan_begin, scan_end;
} jo_scan_layout;

int jo_scan_info(const uint8_t* data, size_t size, int scan_idx, int subseq_bytes, jo_scan_layout* out);

/* Stage twins for one scan; any output pointer may be NULL.
 *   destuffed   [num_subseq * subseq_bytes]   destuffed bytes, every segment zero-padded
 *   seg_offset / seg_count [num_segments], seg_index [num_subseq]
 *   st_*        [num_subseq]  state of a sequential decoder after the last symbol each subsequence
 *               commits (p relative to the segment, n slots, c | z << 8, DC difference sums per scan
 *               component); -1 in p/n/cz for the last subsequence of a segment (not comparable:
 *               it contains the padding)
 *   stream_coef [num_du * 64] */
int jo_scan_stages(
    const uint8_t* data, size_t size, int scan_idx, int subseq_bytes,
    uint8_t* destuffed, int* seg_offset, int* seg_count, int* seg_index,
    int* st_p, int* st_n, int* st_cz, int* st_dc0, int* st_dc1, int* st_dc2, int* st_dc3,
    int16_t* stream_coef);

#ifdef __cplusplus
}
#endif
#endif
