/*
 * jpeg_oracle.c -- CPU restatement of the reference's baseline decode path. TEST INFRASTRUCTURE ONLY:
 * nothing under jpeggpu_amd/ may include, link or call this file; it is the checker that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg compare the HIP path against.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   header / table / geometry parsing      src/reader.cpp:81-672 (status codes, plane sizes :175-181,
 *                                          MCU-rounded data size :395-421), T.81 Annex B
 *   byte rule of destuffing                src/decode_destuff.cu:37-44
 *   symbol semantics                       src/decode_huffman.cu:149-286 (EOB / ZRL / EXTEND), T.81 F.2.2
 *   subsequence ownership rule             src/decode_huffman.cu:302-394 (a symbol that would end past the
 *                                          subsequence's last bit belongs to the next subsequence)
 *   zig-zag order                          src/defs.hpp:94-102
 *   DC de-prediction, int16, per segment   src/decode_dc.cu:42-84,119-163
 *   stream order <-> block raster          src/decode_transpose.cu:65-131
 *   dequant + fixed-point IDCT + clamp     src/idct.cu:44-95,146-223 (every int16 truncation point)
 * Deliberate deviations are those of SURVEY.md Appendix B (B-1 tables persist across scans, B-2
 * non-interleaved scans have one data unit per MCU, B-3 quantiser values are unsigned unless
 * JO_QUIRK_SIGNED_Q is passed).
 *
 * Pinning: the reference as a whole cannot be built in this environment (it needs nvcc and the CUDA runtime
 * headers), and its own tests hold no golden outputs. This oracle is pinned by
 *   (0) the IDCT below == the reference's OWN idct_vector / idct_col / idct_row (src/idct.cu:43-144, compiled from
 *       the reference's source by line range, oracle/ref_lift/build.sh) on 4 672 vectors and 6 144 blocks:
 *       tests/golden/idct_kats.npz, tests/test_idct_kats.py,
 *   (1) quantised coefficients == IJG libjpeg 9d jpeg_read_coefficients on every fixture,
 *   (2) planes within the accuracy band the reference's README states against a standard decoder
 *       (README.md:76,81: MSE 0.15-0.23) -- checked as MSE <= 0.25, max |diff| <= 2 vs libjpeg islow,
 *   (3) the launch shape the reference's README prints for its bundled photo (README.md:37-38:
 *       89 sequences of 256 subsequences of 128 bytes),
 * recorded in tests/golden/ by oracle/pin/make_golden.py and oracle/ref_lift/make_idct_kats.py.
 *
 * Build: gcc -O2 -fwrapv -shared -fPIC (see oracle/Makefile). -fwrapv: the reference's GPU integer
 * arithmetic wraps; signed overflow must not be undefined here.
 */
#include "jpeg_oracle.h"

#include <stdlib.h>
#include <string.h>

static const uint8_t kNatural[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

/* ---------------------------------------------------------------------------------------------- */
/* Huffman tables: T.81 Annex C / F.2.2.3 (MINCODE, MAXCODE, VALPTR), decoded one bit at a time.  */
/* ---------------------------------------------------------------------------------------------- */

typedef struct {
    int defined;
    int mincode[17], maxcode[17], valptr[17]; /* index = code length 1..16 */
    uint8_t huffval[256];
} jo_htab;

static int build_htab(jo_htab* t, const uint8_t bits[16], const uint8_t* vals, int count)
{
    int code = 0, k = 0;
    memset(t, 0, sizeof(*t));
    if (count > 256) return -1;
    memcpy(t->huffval, vals, (size_t)count);
    for (int l = 1; l <= 16; ++l) {
        if (bits[l - 1]) {
            t->valptr[l]  = k;
            t->mincode[l] = code;
            code += bits[l - 1];
            k += bits[l - 1];
            t->maxcode[l] = code - 1;
            if (code > (1 << l)) return -1;
        } else {
            t->maxcode[l] = -1;
        }
        code <<= 1;
    }
    t->defined = 1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------- */
/* Parsed stream                                                                                   */
/* ---------------------------------------------------------------------------------------------- */

typedef struct {
    int comp_idx, dc_id, ac_id, h, v, data_x, data_y;
} jo_scomp;

typedef struct {
    int ncomp;
    jo_scomp c[4];
    size_t begin, end;
    int du_per_mcu, mcus_x, mcus_y, mcus_per_segment;
    jo_htab dc[4], ac[4]; /* tables in force at SOS */
} jo_scan;

typedef struct {
    int width, height, ncomp, hmax, vmax, restart_interval, nscans;
    int id[4], hs[4], vs[4], qidx[4], size_x[4], size_y[4];
    uint16_t qtab[4][64];
    int qdef[4];
    jo_scan scan[4];
} jo_stream;

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
} jo_rd;

static int rd_u8(jo_rd* r) { return *r->p++; }
static int rd_u16(jo_rd* r)
{
    int hi = *r->p++;
    return hi << 8 | *r->p++;
}
static size_t rd_left(const jo_rd* r) { return (size_t)(r->end - r->p); }

static int parse_stream(const uint8_t* data, size_t size, jo_stream* s)
{
    jo_rd r = {data, data + size};
    jo_htab dc[4], ac[4];
    int found_sof = 0, in_scan[4] = {0, 0, 0, 0};
    memset(s, 0, sizeof(*s));
    memset(dc, 0, sizeof(dc));
    memset(ac, 0, sizeof(ac));
    if (rd_left(&r) < 2 || rd_u8(&r) != 0xFF || rd_u8(&r) != 0xD8) return JO_INVALID_JPEG;
    for (;;) {
        int m;
        if (rd_left(&r) < 2) return JO_INVALID_JPEG;
        if (rd_u8(&r) != 0xFF) return JO_INVALID_JPEG;
        m = rd_u8(&r);
        while (m == 0xFF) {
            if (rd_left(&r) < 1) return JO_INVALID_JPEG;
            m = rd_u8(&r);
        }
        if (m == 0xD9) break;
        if (m == 0xC0 || m == 0xC1) { /* SOF0 / SOF1: reference src/reader.cpp:81-184 */
            int len, nc;
            if (found_sof) return JO_INVALID_JPEG;
            found_sof = 1;
            if (rd_left(&r) < 2) return JO_INVALID_JPEG;
            len = rd_u16(&r);
            if (len < 2) return JO_INVALID_JPEG;
            if (rd_left(&r) < (size_t)(len - 2)) return JO_INCOMPLETE;
            if (len < 8) return JO_INVALID_JPEG;
            if (rd_u8(&r) != 8) return JO_NOT_SUPPORTED;
            s->height = rd_u16(&r);
            s->width  = rd_u16(&r);
            if (!s->height || !s->width) return JO_INVALID_JPEG;
            nc = rd_u8(&r);
            if (nc == 0) return JO_INVALID_JPEG;
            if (nc > 4) return JO_NOT_SUPPORTED;
            if (len != 8 + 3 * nc) return JO_INVALID_JPEG;
            s->ncomp = nc;
            for (int c = 0; c < nc; ++c) {
                int sf;
                s->id[c] = rd_u8(&r);
                sf       = rd_u8(&r);
                s->hs[c] = sf >> 4;
                s->vs[c] = sf & 15;
                if (s->hs[c] < 1 || s->hs[c] > 4 || s->vs[c] < 1 || s->vs[c] > 4) return JO_INVALID_JPEG;
                if (nc == 1) s->hs[c] = s->vs[c] = 1; /* :147-153 */
                s->qidx[c] = rd_u8(&r);
                if (s->qidx[c] > 3) return JO_INVALID_JPEG;
                for (int d = 0; d < c; ++d)
                    if (s->id[d] == s->id[c]) return JO_INVALID_JPEG;
                if (s->hs[c] > s->hmax) s->hmax = s->hs[c];
                if (s->vs[c] > s->vmax) s->vmax = s->vs[c];
            }
            for (int c = 0; c < nc; ++c) { /* :175-181 */
                s->size_x[c] = ceil_div(s->width * s->hs[c], s->hmax);
                s->size_y[c] = ceil_div(s->height * s->vs[c], s->vmax);
            }
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return JO_NOT_SUPPORTED; /* :618-630 */
        } else if (m == 0xC4) { /* DHT :226-303 */
            int len, rem;
            if (rd_left(&r) < 2) return JO_INVALID_JPEG;
            len = rd_u16(&r);
            if (len < 2 || rd_left(&r) < (size_t)(len - 2)) return JO_INVALID_JPEG;
            rem = len - 2;
            while (rem > 0) {
                uint8_t bits[16];
                int idx = rd_u8(&r), tc = idx >> 4, th = idx & 15, count = 0;
                --rem;
                if (tc > 1) return JO_INVALID_JPEG;
                if (th > 3) return JO_NOT_SUPPORTED;
                if (rem < 16) return JO_INVALID_JPEG;
                for (int i = 0; i < 16; ++i) {
                    bits[i] = (uint8_t)rd_u8(&r);
                    count += bits[i];
                }
                rem -= 16;
                if (count > 256 || rem < count) return JO_INVALID_JPEG;
                if (build_htab(tc ? &ac[th] : &dc[th], bits, r.p, count)) return JO_INVALID_JPEG;
                r.p += count;
                rem -= count;
            }
        } else if (m == 0xDB) { /* DQT :494-549, stored in natural order :542 */
            int len, rem;
            if (rd_left(&r) < 2) return JO_INVALID_JPEG;
            len = rd_u16(&r);
            if (len < 2 || rd_left(&r) < (size_t)(len - 2)) return JO_INVALID_JPEG;
            rem = len - 2;
            while (rem > 0) {
                int info = rd_u8(&r), prec = info >> 4, id = info & 15, in_use = 0;
                --rem;
                if (prec > 1 || id > 3) return JO_INVALID_JPEG;
                /* 16-bit entries (Pq = 1): the reference stops here (reader.cpp:517-520); libjpeg writes
                   them at low quality, so they are read */
                if (rem < (prec ? 128 : 64)) return JO_INVALID_JPEG;
                for (int c = 0; c < s->ncomp; ++c) in_use |= in_scan[c] && s->qidx[c] == id;
                for (int j = 0; j < 64; ++j) {
                    int q = prec ? rd_u16(&r) : rd_u8(&r);
                    if (!in_use) s->qtab[id][kNatural[j]] = (uint16_t)q;
                }
                s->qdef[id] = 1;
                rem -= prec ? 128 : 64;
            }
        } else if (m == 0xDD) { /* DRI :551-574 */
            int rsti;
            if (rd_left(&r) < 4) return JO_INVALID_JPEG;
            if (rd_u16(&r) != 4) return JO_INVALID_JPEG;
            rsti = rd_u16(&r);
            if (s->nscans > 0 && s->restart_interval != rsti) return JO_NOT_SUPPORTED;
            s->restart_interval = rsti;
        } else if (m == 0xDA) { /* SOS :305-492 */
            int len, ns;
            jo_scan* sc;
            if (!found_sof) return JO_INVALID_JPEG;
            if (rd_left(&r) < 3) return JO_INVALID_JPEG;
            len = rd_u16(&r);
            if (len < 3) return JO_INVALID_JPEG;
            ns = rd_u8(&r);
            if (ns < 1 || ns > 4) return JO_INVALID_JPEG;
            if (s->nscans >= 4) return JO_INVALID_JPEG;
            if (len != 6 + 2 * ns) return JO_INVALID_JPEG;
            if (rd_left(&r) < (size_t)(2 * ns + 3)) return JO_INCOMPLETE;
            sc        = &s->scan[s->nscans];
            sc->ncomp = ns;
            for (int a = 0; a < ns; ++a) {
                int sel = rd_u8(&r), tab = rd_u8(&r), ci = -1, mx, my;
                jo_scomp* q = &sc->c[a];
                for (int i = 0; i < s->ncomp; ++i)
                    if (s->id[i] == sel) {
                        ci = i;
                        break;
                    }
                if (ci < 0) return JO_INVALID_JPEG;
                if (a > 0 && ci <= sc->c[a - 1].comp_idx) return JO_INVALID_JPEG;
                if (in_scan[ci]) return JO_INVALID_JPEG;
                q->comp_idx = ci;
                q->dc_id    = tab >> 4;
                q->ac_id    = tab & 15;
                if (q->dc_id > 3 || q->ac_id > 3) return JO_INVALID_JPEG;
                if (!dc[q->dc_id].defined || !ac[q->ac_id].defined) return JO_INVALID_JPEG;
                if (!s->qdef[s->qidx[ci]]) return JO_INVALID_JPEG;
                q->h      = ns > 1 ? s->hs[ci] : 1; /* Appendix B-2 */
                q->v      = ns > 1 ? s->vs[ci] : 1;
                q->data_x = ceil_div(s->size_x[ci], 8 * q->h) * 8 * q->h;
                q->data_y = ceil_div(s->size_y[ci], 8 * q->v) * 8 * q->v;
                mx        = q->data_x / (8 * q->h);
                my        = q->data_y / (8 * q->v);
                if (a > 0 && (mx != sc->mcus_x || my != sc->mcus_y)) return JO_NOT_SUPPORTED;
                sc->mcus_x = mx;
                sc->mcus_y = my;
                sc->du_per_mcu += q->h * q->v;
            }
            if (sc->du_per_mcu > 10) return JO_INVALID_JPEG;
            for (int a = 0; a < ns; ++a) in_scan[sc->c[a].comp_idx] = 1;
            r.p += 3;
            memcpy(sc->dc, dc, sizeof(dc));
            memcpy(sc->ac, ac, sizeof(ac));
            sc->mcus_per_segment = s->restart_interval ? s->restart_interval : sc->mcus_x * sc->mcus_y;
            sc->begin            = (size_t)(r.p - data);
            ++s->nscans;
            /* skip entropy-coded data: up to the first marker that is not RSTn (:447-489) */
            for (;;) {
                const uint8_t* q = memchr(r.p, 0xFF, rd_left(&r));
                if (!q || q + 1 >= r.end) return JO_INVALID_JPEG;
                if (q[1] == 0 || (q[1] >= 0xD0 && q[1] <= 0xD7)) {
                    r.p = q + 2;
                    continue;
                }
                if (q[1] == 0xFF) {
                    r.p = q + 1;
                    continue;
                }
                /* fill bytes before this marker belong to it: back up over them */
                while (q > data + sc->begin && q[-1] == 0xFF) --q;
                r.p     = q;
                sc->end = (size_t)(q - data);
                break;
            }
        } else if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) {
            return JO_INVALID_JPEG;
        } else { /* skipped segment :576-594 */
            int len;
            if (rd_left(&r) < 2) return JO_INVALID_JPEG;
            len = rd_u16(&r);
            if (len < 2) return JO_INVALID_JPEG;
            if (rd_left(&r) < (size_t)(len - 2)) return JO_INCOMPLETE;
            r.p += len - 2;
        }
    }
    if (!found_sof || s->nscans == 0) return JO_INVALID_JPEG;
    for (int c = 0; c < s->ncomp; ++c)
        if (!in_scan[c]) return JO_INVALID_JPEG;
    return JO_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* Destuffed, segment-padded layout of one scan (reference src/decode_destuff.cu)                  */
/* ---------------------------------------------------------------------------------------------- */

typedef struct {
    uint8_t* bytes;    /* num_subseq * subseq_bytes, each segment zero-padded */
    int* seg_offset;   /* subsequences before segment i */
    int* seg_count;    /* subsequences in segment i */
    int* seg_bytes;    /* data bytes in segment i */
    int num_segments, num_subseq;
} jo_layout;

static void layout_free(jo_layout* l)
{
    free(l->bytes);
    free(l->seg_offset);
    free(l->seg_count);
    free(l->seg_bytes);
    memset(l, 0, sizeof(*l));
}

/* byte rule, reference src/decode_destuff.cu:37-44 (is_byte_data, called with prev_is_stuffing = the byte in front is
   FF, :64 and :91): a byte is data iff it is the 00 behind an FF (it then stands for FF) or neither it nor the byte in front
   of it is FF */
static int byte_rule(int prev, int b, uint8_t* write)
{
    if (prev == 0xFF && b == 0x00) {
        *write = 0xFF;
        return 1;
    }
    if (prev != 0xFF && b != 0xFF) {
        *write = (uint8_t)b;
        return 1;
    }
    return 0;
}

void jo_byte_rule(const uint8_t* prev, const uint8_t* byte, int n, uint8_t* is_data, uint8_t* written)
{
    for (int i = 0; i < n; ++i) {
        uint8_t w   = 0;
        is_data[i] = (uint8_t)byte_rule(prev[i], byte[i], &w);
        written[i] = w;
    }
}

static int build_layout(const uint8_t* data, const jo_scan* sc, int subseq_bytes, jo_layout* l)
{
    const uint8_t* p   = data + sc->begin;
    const uint8_t* end = data + sc->end;
    size_t cap_b = (size_t)(sc->end - sc->begin) + 2 * (size_t)subseq_bytes, nb = 0;
    int cap_s = 16, ns = 0, prev = 0, seg_start_b = 0;
    memset(l, 0, sizeof(*l));
    /* every segment may add up to subseq_bytes - 1 bytes of padding: count markers first */
    for (const uint8_t* q = p; q + 1 < end; ++q)
        if (q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7) cap_b += (size_t)subseq_bytes;
    l->bytes      = calloc(cap_b, 1);
    l->seg_offset = malloc(sizeof(int) * (size_t)cap_s);
    l->seg_count  = malloc(sizeof(int) * (size_t)cap_s);
    l->seg_bytes  = malloc(sizeof(int) * (size_t)cap_s);
    if (!l->bytes || !l->seg_offset || !l->seg_count || !l->seg_bytes) return JO_NO_MEMORY;
    for (;; ++p) {
        int at_end = p >= end, is_rst = 0;
        if (!at_end) {
            int b = *p;
            uint8_t w;
            if (byte_rule(prev, b, &w)) l->bytes[nb++] = w;
            else if (prev == 0xFF && b >= 0xD0 && b <= 0xD7) is_rst = 1;
            prev = is_rst ? 0 : b;
        }
        if (at_end || is_rst) {
            int bytes = (int)nb - seg_start_b, cnt = ceil_div(bytes, subseq_bytes);
            /* a restart interval holds at least one MCU: a segment without a byte (two markers back to back, which
               the reference's walk src/reader.cpp:447-489 would count as a segment of no subsequences) is refused */
            if (bytes == 0) return JO_INVALID_JPEG;
            if (ns == cap_s) {
                cap_s *= 2;
                l->seg_offset = realloc(l->seg_offset, sizeof(int) * (size_t)cap_s);
                l->seg_count  = realloc(l->seg_count, sizeof(int) * (size_t)cap_s);
                l->seg_bytes  = realloc(l->seg_bytes, sizeof(int) * (size_t)cap_s);
                if (!l->seg_offset || !l->seg_count || !l->seg_bytes) return JO_NO_MEMORY;
            }
            l->seg_offset[ns] = l->num_subseq;
            l->seg_count[ns]  = cnt;
            l->seg_bytes[ns]  = bytes;
            ++ns;
            l->num_subseq += cnt;
            nb          = (size_t)l->num_subseq * (size_t)subseq_bytes; /* zero padding (calloc) */
            seg_start_b = (int)nb;
            if (at_end) break;
        }
    }
    l->num_segments = ns;
    return JO_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* Sequential entropy decode of one scan into the stream-order buffer                              */
/* ---------------------------------------------------------------------------------------------- */

typedef struct {
    const uint8_t* b;
    int nbits; /* bits available in this segment's data */
    int pos;
} jo_bits;

static int get_bit(jo_bits* br)
{
    int v = 0;
    if (br->pos < br->nbits) v = (br->b[br->pos >> 3] >> (7 - (br->pos & 7))) & 1;
    ++br->pos; /* zero-extended past the end */
    return v;
}

static int decode_sym(jo_bits* br, const jo_htab* t)
{
    /* T.81 F.2.2.3 DECODE; a 16-bit prefix that matches nothing is taken as length 16 with the
       huffval index reduced modulo 256 (reference src/decode_huffman.cu:177-193) */
    int code = 0, l;
    for (l = 1; l <= 16; ++l) {
        code = code << 1 | get_bit(br);
        if (code <= t->maxcode[l]) break; /* maxcode is -1 for unused lengths */
    }
    if (l > 16) l = 16;
    return t->huffval[(uint8_t)(t->valptr[l] + code - t->mincode[l])];
}

static int receive_extend(jo_bits* br, int s)
{
    int v = 0;
    for (int i = 0; i < s; ++i) v = v << 1 | get_bit(br);
    if (s && v < (1 << (s - 1))) v = v - (1 << s) + 1; /* T.81 F.2.2.1 */
    return v;
}

/* One symbol per 32-bit window (most significant bit first, zeros behind it), the outputs of the reference's
   decode_next_symbol<true> (src/decode_huffman.cu:202-286): bits taken, coefficient value, run of zeros in front of
   it (AC: 15 for ZRL, 63 - z for an end of block). For the tests against the reference-built known answers
   (tests/golden/huff_kats.npz). */
int jo_symbol_steps(const uint8_t bits[16], const uint8_t* vals, int count, int is_dc, const uint32_t* win, const int* z, int n,
                    int* length, int* symbol, int* run)
{
    jo_htab t;
    if (build_htab(&t, bits, vals, count)) return JO_INVALID_JPEG;
    for (int i = 0; i < n; ++i) {
        uint8_t b[4] = {(uint8_t)(win[i] >> 24), (uint8_t)(win[i] >> 16), (uint8_t)(win[i] >> 8), (uint8_t)win[i]};
        jo_bits br  = {b, 32, 0};
        int sym = decode_sym(&br, &t), ssss = sym & 15, rrrr = is_dc ? 0 : sym >> 4;
        symbol[i]   = receive_extend(&br, ssss);
        length[i]   = br.pos;
        run[i]      = is_dc ? 0 : ssss ? rrrr : rrrr == 15 ? 15 : 63 - z[i];
    }
    return JO_OK;
}

/* Optional record of the decoder state at subsequence boundaries (stage twin of the sync passes). */
typedef struct {
    int subseq_bits;
    int* p;
    int* n;
    int* cz;
    int* dc[4];
} jo_states;

static int decode_scan(
    const jo_stream* s, const jo_scan* sc, const jo_layout* l, int subseq_bytes, int16_t* out, jo_states* rec)
{
    const int total_mcus = sc->mcus_x * sc->mcus_y;
    int du_comp[10], ndu = 0;
    (void)s;
    for (int a = 0; a < sc->ncomp; ++a)
        for (int k = 0; k < sc->c[a].h * sc->c[a].v; ++k) du_comp[ndu++] = a;
    if (l->num_segments != ceil_div(total_mcus, sc->mcus_per_segment)) return JO_INVALID_JPEG;
    memset(out, 0, sizeof(int16_t) * (size_t)total_mcus * (size_t)ndu * 64);

    for (int seg = 0; seg < l->num_segments; ++seg) {
        jo_bits br = {l->bytes + (size_t)l->seg_offset[seg] * (size_t)subseq_bytes, l->seg_bytes[seg] * 8, 0};
        int mcu0 = seg * sc->mcus_per_segment, mcu1 = mcu0 + sc->mcus_per_segment;
        int pred[4] = {0, 0, 0, 0};
        /* running record for the stage twin */
        int cur_sub = 0, n_in_sub = 0, dc_in_sub[4] = {0, 0, 0, 0};
        if (mcu1 > total_mcus) mcu1 = total_mcus; /* Appendix B-5 */
        for (int mcu = mcu0; mcu < mcu1; ++mcu) {
            for (int c = 0; c < ndu; ++c) {
                const jo_scomp* q = &sc->c[du_comp[c]];
                int16_t* blk      = out + ((size_t)mcu * (size_t)ndu + (size_t)c) * 64;
                int z             = 0;
                while (z < 64) {
                    int start = br.pos, adv, diff = 0, sym, ssss, rrrr;
                    (void)start;
                    if (z == 0) {
                        sym  = decode_sym(&br, &sc->dc[q->dc_id]);
                        ssss = sym & 15;
                        diff = receive_extend(&br, ssss);
                        pred[du_comp[c]] += diff;
                        blk[0] = (int16_t)pred[du_comp[c]]; /* int16 like decode_dc.cu:129-155 */
                        adv    = 1;
                    } else {
                        sym  = decode_sym(&br, &sc->ac[q->ac_id]);
                        ssss = sym & 15;
                        rrrr = sym >> 4;
                        if (ssss == 0) {
                            adv = rrrr == 15 ? 16 : 64 - z; /* ZRL / EOB, decode_huffman.cu:241-247 */
                        } else {
                            int v = receive_extend(&br, ssss);
                            if (z + rrrr < 64) blk[kNatural[z + rrrr]] = (int16_t)v;
                            adv = rrrr + 1;
                        }
                    }
                    if (rec) {
                        /* the symbol [start, br.pos) is committed by the subsequence in which it ENDS */
                        int owner = (br.pos + rec->subseq_bits - 1) / rec->subseq_bits - 1;
                        while (cur_sub < owner) { /* close subsequences that ended before this symbol */
                            int g = l->seg_offset[seg] + cur_sub;
                            rec->p[g]  = start;
                            rec->n[g]  = n_in_sub;
                            rec->cz[g] = c | z << 8;
                            for (int k = 0; k < 4; ++k)
                                if (rec->dc[k]) rec->dc[k][g] = dc_in_sub[k];
                            n_in_sub = 0;
                            memset(dc_in_sub, 0, sizeof(dc_in_sub));
                            ++cur_sub;
                        }
                        n_in_sub += adv;
                        if (z == 0) dc_in_sub[du_comp[c]] += diff;
                    }
                    z += adv;
                }
            }
        }
        if (br.pos > br.nbits) return JO_INVALID_JPEG; /* ran past the segment's data */
        if (rec) {
            /* subsequences from cur_sub on contain the end of the data: state not comparable */
            for (int k = cur_sub; k < l->seg_count[seg]; ++k) {
                int g      = l->seg_offset[seg] + k;
                rec->p[g]  = -1;
                rec->n[g]  = -1;
                rec->cz[g] = -1;
            }
        }
    }
    return JO_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* dequant + IDCT, restating reference src/idct.cu:44-95 and :146-223                              */
/* ---------------------------------------------------------------------------------------------- */

static int unfixh(int x) { return (int16_t)((x + 0x8000) >> 16); }
static int unfixo(int x) { return (x + 0x1000) >> 13; }

static void idct_vector(int* v0, int* v1, int* v2, int* v3, int* v4, int* v5, int* v6, int* v7)
{
    const int cos_1_4 = 0x5a82, sin_1_8 = 0x30fc, cos_1_8 = 0x7642;
    const int osin_1_16 = 0x063e, osin_5_16 = 0x1a9b, ocos_1_16 = 0x1f63, ocos_5_16 = 0x11c7;
    int tmp10 = (*v0 + *v4) * cos_1_4;
    int tmp11 = (*v0 - *v4) * cos_1_4;
    int tmp12 = *v2 * sin_1_8 - *v6 * cos_1_8;
    int tmp13 = *v6 * sin_1_8 + *v2 * cos_1_8;
    int tmp20 = tmp10 + tmp13, tmp21 = tmp11 + tmp12, tmp22 = tmp11 - tmp12, tmp23 = tmp10 - tmp13;
    int tmp30 = unfixo((*v3 + *v5) * cos_1_4);
    int tmp31 = unfixo((*v3 - *v5) * cos_1_4);
    int w1 = (int)((unsigned)*v1 << 2), w7 = (int)((unsigned)*v7 << 2);
    int tmp40 = w1 + tmp30, tmp41 = w7 + tmp31, tmp42 = w1 - tmp30, tmp43 = w7 - tmp31;
    int tmp50 = tmp40 * ocos_1_16 + tmp41 * osin_1_16;
    int tmp51 = tmp40 * osin_1_16 - tmp41 * ocos_1_16;
    int tmp52 = tmp42 * ocos_5_16 + tmp43 * osin_5_16;
    int tmp53 = tmp42 * osin_5_16 - tmp43 * ocos_5_16;
    *v0 = unfixh(tmp20 + tmp50);
    *v1 = unfixh(tmp21 + tmp53);
    *v2 = unfixh(tmp22 + tmp52);
    *v3 = unfixh(tmp23 + tmp51);
    *v4 = unfixh(tmp23 - tmp51);
    *v5 = unfixh(tmp22 - tmp52);
    *v6 = unfixh(tmp21 - tmp53);
    *v7 = unfixh(tmp20 - tmp50);
}

/* the 8-point transform alone on n vectors (pinned by tests/golden/idct_kats.npz: vec_in -> vec_out) */
void jo_idct_vectors(const int32_t* in, int32_t* out, int n)
{
    for (int i = 0; i < n; ++i) {
        int v[8];
        for (int k = 0; k < 8; ++k) v[k] = in[8 * i + k];
        idct_vector(&v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
        for (int k = 0; k < 8; ++k) out[8 * i + k] = v[k];
    }
}

void jo_idct_block(const int16_t coef[64], const uint16_t q[64], uint8_t out[64], int flags)
{
    int16_t blk[64];
    for (int i = 0; i < 64; ++i) { /* idct.cu:176-180: int16 = int16 * (int8 | uint8) */
        int qv = (flags & JO_QUIRK_SIGNED_Q) ? (int)(int8_t)(uint8_t)q[i] : (int)q[i];
        blk[i] = (int16_t)(coef[i] * qv);
    }
    for (int x = 0; x < 8; ++x) { /* column pass, idct.cu:97-120 */
        int v[8];
        for (int i = 0; i < 8; ++i) v[i] = blk[i * 8 + x];
        idct_vector(&v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
        for (int i = 0; i < 8; ++i) blk[i * 8 + x] = (int16_t)v[i];
    }
    for (int y = 0; y < 8; ++y) { /* row pass, idct.cu:122-144 */
        int v[8];
        for (int i = 0; i < 8; ++i) v[i] = blk[y * 8 + i];
        idct_vector(&v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
        for (int i = 0; i < 8; ++i) {
            int16_t val    = (int16_t)(v[i] + 128); /* idct.cu:218-220 */
            out[y * 8 + i] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
        }
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* Public entry points                                                                              */
/* ---------------------------------------------------------------------------------------------- */

void jo_free(jo_image* img)
{
    for (int c = 0; c < 4; ++c) {
        free(img->coef[c]);
        free(img->plane[c]);
    }
    for (int i = 0; i < 4; ++i) free(img->stream_coef[i]);
    memset(img, 0, sizeof(*img));
}

int jo_decode(const uint8_t* data, size_t size, jo_image* img, int flags)
{
    jo_stream s;
    int rc = parse_stream(data, size, &s);
    memset(img, 0, sizeof(*img));
    if (rc) return rc;
    img->width            = s.width;
    img->height           = s.height;
    img->ncomp            = s.ncomp;
    img->restart_interval = s.restart_interval;
    img->nscans           = s.nscans;
    memcpy(img->qtab, s.qtab, sizeof(s.qtab));
    for (int c = 0; c < s.ncomp; ++c) {
        img->hs[c]      = s.hs[c];
        img->vs[c]      = s.vs[c];
        img->qidx[c]    = s.qidx[c];
        img->plane_w[c] = s.size_x[c];
        img->plane_h[c] = s.size_y[c];
    }
    for (int i = 0; i < s.nscans && !rc; ++i) {
        const jo_scan* sc = &s.scan[i];
        jo_layout l;
        const size_t ndu = (size_t)sc->mcus_x * sc->mcus_y * sc->du_per_mcu;
        int16_t* so      = malloc(sizeof(int16_t) * ndu * 64 + 2);
        if (!so) {
            rc = JO_NO_MEMORY;
            break;
        }
        img->stream_coef[i]   = so;
        img->stream_du[i]     = (int)ndu;
        img->scan_ncomp[i]    = sc->ncomp;
        img->scan_du_per_mcu[i] = sc->du_per_mcu;
        rc = build_layout(data, sc, 128, &l);
        if (!rc) rc = decode_scan(&s, sc, &l, 128, so, NULL);
        layout_free(&l);
        if (rc) break;
        /* stream order -> block raster per component (reference decode_transpose.cu:65-131) + IDCT */
        {
            int du = 0;
            for (int a = 0; a < sc->ncomp && !rc; ++a) {
                const jo_scomp* q = &sc->c[a];
                const int ci = q->comp_idx, bw = q->data_x / 8, bh = q->data_y / 8;
                img->scan_comp[i][a] = ci;
                img->blocks_w[ci]    = bw;
                img->blocks_h[ci]    = bh;
                img->coef[ci]        = malloc(sizeof(int16_t) * (size_t)bw * bh * 64);
                img->plane[ci]       = malloc((size_t)s.size_x[ci] * s.size_y[ci]);
                if (!img->coef[ci] || !img->plane[ci]) {
                    rc = JO_NO_MEMORY;
                    break;
                }
                for (int my = 0; my < sc->mcus_y; ++my)
                    for (int mx = 0; mx < sc->mcus_x; ++mx)
                        for (int dy = 0; dy < q->v; ++dy)
                            for (int dx = 0; dx < q->h; ++dx) {
                                const size_t mcu = (size_t)my * sc->mcus_x + mx;
                                const int16_t* src =
                                    so + (mcu * sc->du_per_mcu + (size_t)(du + dy * q->h + dx)) * 64;
                                const int bx = mx * q->h + dx, by = my * q->v + dy;
                                uint8_t px[64];
                                memcpy(img->coef[ci] + ((size_t)by * bw + bx) * 64, src, 128);
                                jo_idct_block(src, s.qtab[s.qidx[ci]], px, flags);
                                for (int y = 0; y < 8; ++y) {
                                    const int yy = by * 8 + y;
                                    if (yy >= s.size_y[ci]) break;
                                    for (int x = 0; x < 8; ++x) {
                                        const int xx = bx * 8 + x;
                                        if (xx >= s.size_x[ci]) break;
                                        img->plane[ci][(size_t)yy * s.size_x[ci] + xx] = px[y * 8 + x];
                                    }
                                }
                            }
                du += q->h * q->v;
            }
        }
    }
    if (rc) jo_free(img);
    return rc;
}

int jo_scan_info(const uint8_t* data, size_t size, int scan_idx, int subseq_bytes, jo_scan_layout* out)
{
    jo_stream s;
    jo_layout l;
    int rc = parse_stream(data, size, &s);
    memset(out, 0, sizeof(*out));
    if (rc) return rc;
    if (scan_idx < 0 || scan_idx >= s.nscans) return JO_INVALID_ARGUMENT;
    rc = build_layout(data, &s.scan[scan_idx], subseq_bytes, &l);
    if (rc) return rc;
    out->num_subseq   = l.num_subseq;
    out->num_segments = l.num_segments;
    out->num_du       = s.scan[scan_idx].mcus_x * s.scan[scan_idx].mcus_y * s.scan[scan_idx].du_per_mcu;
    out->scan_begin   = s.scan[scan_idx].begin;
    out->scan_end     = s.scan[scan_idx].end;
    layout_free(&l);
    return JO_OK;
}

int jo_scan_stages(
    const uint8_t* data,
    size_t size,
    int scan_idx,
    int subseq_bytes,
    uint8_t* destuffed,
    int* seg_offset,
    int* seg_count,
    int* seg_index,
    int* st_p,
    int* st_n,
    int* st_cz,
    int* st_dc0,
    int* st_dc1,
    int* st_dc2,
    int* st_dc3,
    int16_t* stream_coef)
{
    jo_stream s;
    jo_layout l;
    jo_states rec;
    int rc = parse_stream(data, size, &s);
    if (rc) return rc;
    if (scan_idx < 0 || scan_idx >= s.nscans) return JO_INVALID_ARGUMENT;
    rc = build_layout(data, &s.scan[scan_idx], subseq_bytes, &l);
    if (rc) return rc;
    if (destuffed) memcpy(destuffed, l.bytes, (size_t)l.num_subseq * (size_t)subseq_bytes);
    for (int i = 0; i < l.num_segments; ++i) {
        if (seg_offset) seg_offset[i] = l.seg_offset[i];
        if (seg_count) seg_count[i] = l.seg_count[i];
        if (seg_index)
            for (int k = 0; k < l.seg_count[i]; ++k) seg_index[l.seg_offset[i] + k] = i;
    }
    rec.subseq_bits = subseq_bytes * 8;
    rec.p           = st_p;
    rec.n           = st_n;
    rec.cz          = st_cz;
    rec.dc[0]       = st_dc0;
    rec.dc[1]       = st_dc1;
    rec.dc[2]       = st_dc2;
    rec.dc[3]       = st_dc3;
    if (stream_coef) rc = decode_scan(&s, &s.scan[scan_idx], &l, subseq_bytes, stream_coef, st_p ? &rec : NULL);
    layout_free(&l);
    return rc;
}
