/* ijg_dump.c -- pin tooling (runs only in the authoring container, needs IJG libjpeg 9d from
 * /opt/conda): dumps what an independent, standard decoder produces for a JPEG file:
 *   mode "coef":  quantised coefficients via jpeg_read_coefficients, per component
 *                 int32 blocks_w, int32 blocks_h, then int16[blocks_h][blocks_w][64] (natural order)
 *   mode "raw":   raw (non-upsampled, non-colour-converted) planes with JDCT_ISLOW, per component
 *                 int32 w, int32 h (cropped to ceil(W*h/hmax)), then uint8[h][w]
 * Used by make_golden.py to pin the oracle (coefficients exact; planes within the accuracy band of
 * the reference's README.md:76,81). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <jpeglib.h>

static void wr(const void* p, size_t n, FILE* f) { if (fwrite(p, 1, n, f) != n) exit(3); }

int main(int argc, char** argv)
{
    if (argc != 4) { fprintf(stderr, "usage: ijg_dump coef|raw in.jpg out.bin\n"); return 2; }
    FILE* in = fopen(argv[2], "rb"); FILE* out = fopen(argv[3], "wb");
    if (!in || !out) return 2;
    struct jpeg_decompress_struct c; struct jpeg_error_mgr e;
    c.err = jpeg_std_error(&e); jpeg_create_decompress(&c); jpeg_stdio_src(&c, in);
    jpeg_read_header(&c, TRUE);
    if (!strcmp(argv[1], "coef")) {
        jvirt_barray_ptr* arrs = jpeg_read_coefficients(&c);
        int nc = c.num_components; wr(&nc, 4, out);
        for (int ci = 0; ci < nc; ++ci) {
            jpeg_component_info* comp = &c.comp_info[ci];
            int bw = comp->width_in_blocks, bh = comp->height_in_blocks;
            wr(&bw, 4, out); wr(&bh, 4, out);
            for (int by = 0; by < bh; ++by) {
                JBLOCKARRAY rows = c.mem->access_virt_barray((j_common_ptr)&c, arrs[ci], by, 1, FALSE);
                for (int bx = 0; bx < bw; ++bx) {
                    short blk[64];
                    for (int k = 0; k < 64; ++k) blk[k] = rows[0][bx][k];
                    wr(blk, 128, out);
                }
            }
        }
    } else {
        c.raw_data_out = TRUE; c.dct_method = JDCT_ISLOW; c.do_fancy_upsampling = FALSE;
        jpeg_start_decompress(&c);
        int nc = c.num_components; wr(&nc, 4, out);
        int maxv = c.max_v_samp_factor, maxh = c.max_h_samp_factor;
        int lines = maxv * c.min_DCT_v_scaled_size;
        unsigned char** planes = calloc(nc, sizeof(*planes)); int pw[4], ph[4], fw[4];
        JSAMPARRAY arr[4];
        for (int ci = 0; ci < nc; ++ci) {
            jpeg_component_info* comp = &c.comp_info[ci];
            fw[ci] = comp->width_in_blocks * 8;
            int fh = ((c.image_height + lines - 1) / lines) * comp->v_samp_factor * 8 + 64;
            planes[ci] = calloc((size_t)fw[ci] * fh, 1);
            pw[ci] = (c.image_width * comp->h_samp_factor + maxh - 1) / maxh;
            ph[ci] = (c.image_height * comp->v_samp_factor + maxv - 1) / maxv;
            arr[ci] = malloc(sizeof(JSAMPROW) * comp->v_samp_factor * 8);
        }
        int row = 0;
        while (c.output_scanline < c.output_height) {
            for (int ci = 0; ci < nc; ++ci) {
                int v = c.comp_info[ci].v_samp_factor * 8;
                for (int r = 0; r < v; ++r) arr[ci][r] = planes[ci] + (size_t)(row * v + r) * fw[ci];
            }
            jpeg_read_raw_data(&c, arr, lines); ++row;
        }
        for (int ci = 0; ci < nc; ++ci) {
            wr(&pw[ci], 4, out); wr(&ph[ci], 4, out);
            for (int y = 0; y < ph[ci]; ++y) wr(planes[ci] + (size_t)y * fw[ci], pw[ci], out);
        }
        jpeg_finish_decompress(&c);
    }
    jpeg_destroy_decompress(&c); fclose(in); fclose(out); return 0;
}
