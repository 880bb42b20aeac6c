/*
 * jpegsynth.c -- seeded synthetic baseline-JPEG generator (test / benchmark tooling, not product).
 *
 * Neither the reference's test images (external repository) nor a network exist on the GPU box, and
 * neither cjpeg nor Pillow can produce every BASELINE.json configuration (non-interleaved scans,
 * 4 components with 4+4 Huffman tables, arbitrary sampling factors, arbitrary restart intervals).
 * This encoder paints a procedural image (gradients + sinusoids + two octaves of value noise + white
 * noise), subsamples it by box filter, runs a float FDCT, quantises with scaled Annex-K tables
 * (entries clamped to 1..127) and Huffman-codes it with either the Annex-K
 * tables or per-component optimised tables (T.81 K.2). DHT segments are re-emitted before every SOS
 * so the files are also valid input for the reference (Appendix B-1).
 *
 * Build: gcc -O2 -shared -fPIC -o libjpegsynth.so jpegsynth.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int width, height, ncomp;
    int hs[4], vs[4];
    int interleaved;      /* 1: one scan with all components; 0: one scan per component */
    int restart_interval; /* MCUs, 0 = none */
    int quality;          /* 1..100, libjpeg-style scaling of the Annex-K quantisation tables */
    int optimize;         /* 0: Annex-K Huffman tables (luma for component 0, chroma otherwise);
                             1: one optimised DC+AC pair per component (table id = component index) */
    int noise;            /* amplitude of the white-noise term, 0..64 */
    int fill_bytes;       /* number of FF fill bytes inserted before every RSTn / EOI marker */
    uint64_t seed;
    int qmax;             /* largest quantiser value; 0 = 127. Above 255 the tables are written with 16-bit
                             entries (Pq = 1), as libjpeg does at low quality without -baseline */
} js_params;

static const uint8_t kZigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* T.81 Annex K.1 */
static const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                   14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                   18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                     24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
/* T.81 Annex K.3 */
static const uint8_t kDcLumaBits[16]   = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcVals[12]       = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcLumaBits[16]   = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kAcLumaVals[162]  = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
    uint8_t bits[16];
    uint8_t vals[256];
    int count;
    uint16_t code[256];
    uint8_t size[256];
} htab;

static void htab_finish(htab* t)
{
    int k = 0, code = 0;
    memset(t->code, 0, sizeof(t->code));
    memset(t->size, 0, sizeof(t->size));
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < t->bits[l - 1]; ++i, ++k) {
            t->code[t->vals[k]] = (uint16_t)code++;
            t->size[t->vals[k]] = (uint8_t)l;
        }
        code <<= 1;
    }
    t->count = k;
}

static void htab_std(htab* t, const uint8_t* bits, const uint8_t* vals, int n)
{
    memset(t, 0, sizeof(*t));
    memcpy(t->bits, bits, 16);
    memcpy(t->vals, vals, (size_t)n);
    htab_finish(t);
}

/* T.81 K.2: optimal code lengths limited to 16 bits, symbol 256 reserved so no code is all ones */
static void htab_optimal(htab* t, const long* freq_in)
{
    long freq[257];
    int codesize[257], others[257];
    uint8_t bits[33];
    memcpy(freq, freq_in, sizeof(long) * 256);
    freq[256] = 1;
    memset(codesize, 0, sizeof(codesize));
    memset(bits, 0, sizeof(bits));
    for (int i = 0; i < 257; ++i) others[i] = -1;
    for (;;) {
        int c1 = -1, c2 = -1;
        long v = 1000000000L;
        for (int i = 0; i <= 256; ++i)
            if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; ++i)
            if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2];
        freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i <= 256; ++i)
        if (codesize[i]) bits[codesize[i] > 32 ? 32 : codesize[i]]++;
    for (int i = 32; i > 16; --i) {
        while (bits[i] > 0) {
            int j = i - 2;
            while (bits[j] == 0) --j;
            bits[i] -= 2;
            bits[i - 1]++;
            bits[j + 1] += 2;
            bits[j]--;
        }
    }
    {
        int i = 16;
        while (bits[i] == 0) --i;
        bits[i]--; /* remove the reserved symbol */
    }
    memset(t, 0, sizeof(*t));
    memcpy(t->bits, bits + 1, 16);
    {
        int k = 0;
        for (int l = 1; l <= 32; ++l)
            for (int i = 0; i < 256; ++i)
                if (codesize[i] == l) t->vals[k++] = (uint8_t)i;
    }
    htab_finish(t);
}

/* ---- output with byte stuffing ---- */
typedef struct {
    uint8_t* out;
    size_t n, cap;
    uint32_t acc;
    int nbits;
    int overflow;
} bitw;

static void put_byte_raw(bitw* w, int b)
{
    if (w->n < w->cap) w->out[w->n++] = (uint8_t)b;
    else w->overflow = 1;
}
static void put_bits(bitw* w, unsigned code, int size)
{
    if (!size) return;
    w->acc = (w->acc << size) | (code & ((1u << size) - 1));
    w->nbits += size;
    while (w->nbits >= 8) {
        int b = (w->acc >> (w->nbits - 8)) & 0xFF;
        put_byte_raw(w, b);
        if (b == 0xFF) put_byte_raw(w, 0);
        w->nbits -= 8;
    }
}
static void flush_bits(bitw* w)
{
    if (w->nbits) put_bits(w, 0x7F, 8 - w->nbits); /* pad with ones */
    w->acc   = 0;
    w->nbits = 0;
}
static void put_marker(bitw* w, int m)
{
    put_byte_raw(w, 0xFF);
    put_byte_raw(w, m);
}
static void put_u16(bitw* w, int v)
{
    put_byte_raw(w, v >> 8);
    put_byte_raw(w, v & 0xFF);
}

static int nbits_of(int v)
{
    int n = 0;
    if (v < 0) v = -v;
    while (v) { ++n; v >>= 1; }
    return n;
}

/* encode (or, with w == NULL, count symbol statistics of) one block; coef in natural order */
static void code_block(bitw* w, const int16_t* blk, int* pred, const htab* dc, const htab* ac, long* fdc, long* fac)
{
    int diff = blk[0] - *pred, s = nbits_of(diff), run = 0;
    *pred = blk[0];
    if (w) {
        put_bits(w, dc->code[s], dc->size[s]);
        if (s) put_bits(w, (unsigned)(diff < 0 ? diff - 1 : diff), s);
    } else {
        fdc[s]++;
    }
    for (int k = 1; k < 64; ++k) {
        int v = blk[kZigzag[k]];
        if (v == 0) { ++run; continue; }
        while (run > 15) {
            if (w) put_bits(w, ac->code[0xF0], ac->size[0xF0]);
            else fac[0xF0]++;
            run -= 16;
        }
        s = nbits_of(v);
        if (w) {
            put_bits(w, ac->code[run << 4 | s], ac->size[run << 4 | s]);
            put_bits(w, (unsigned)(v < 0 ? v - 1 : v), s);
        } else {
            fac[run << 4 | s]++;
        }
        run = 0;
    }
    if (run) {
        if (w) put_bits(w, ac->code[0], ac->size[0]);
        else fac[0]++;
    }
}

/* ---- procedural image ---- */
static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static float lattice(uint64_t seed, int ix, int iy)
{
    return (float)(mix64(seed ^ ((uint64_t)(uint32_t)ix << 32 | (uint32_t)iy)) >> 40) * (1.0f / 16777216.0f);
}
static float value_noise(uint64_t seed, float x, float y)
{
    int ix = (int)floorf(x), iy = (int)floorf(y);
    float fx = x - ix, fy = y - iy;
    float a = lattice(seed, ix, iy), b = lattice(seed, ix + 1, iy), c = lattice(seed, ix, iy + 1), d = lattice(seed, ix + 1, iy + 1);
    fx = fx * fx * (3 - 2 * fx);
    fy = fy * fy * (3 - 2 * fy);
    return a + (b - a) * fx + (c - a) * fy + (a - b - c + d) * fx * fy;
}

static void paint_row(const js_params* p, int comp, int y, float* row /* width */)
{
    const uint64_t s = p->seed * 0x9E3779B97F4A7C15ULL + (uint64_t)comp * 0x1234567ULL;
    const float amp  = comp == 0 ? 1.0f : 0.45f;
    const float ph   = (float)(mix64(s) & 1023) * 0.00613f;
    for (int x = 0; x < p->width; ++x) {
        float v = 128.0f;
        v += amp * 40.0f * sinf(x * 0.0131f + y * 0.0077f + ph);
        v += amp * 22.0f * sinf(x * 0.071f - y * 0.049f + 2 * ph);
        v += amp * 70.0f * (value_noise(s, x * (1.0f / 96), y * (1.0f / 96)) - 0.5f);
        v += amp * 36.0f * (value_noise(s + 7, x * (1.0f / 11), y * (1.0f / 11)) - 0.5f);
        /* edges: a coarse checker whose cells flip brightness */
        v += amp * ((mix64(s ^ (uint64_t)((x >> 6) * 7919 + (y >> 6))) & 3) == 0 ? 28.0f : 0.0f);
        if (p->noise) v += amp * (float)p->noise * ((float)(mix64(s ^ ((uint64_t)y << 32 | (uint32_t)x)) >> 40) * (1.0f / 8388608.0f) - 1.0f);
        row[x] = v < 0 ? 0 : v > 255 ? 255 : v;
    }
}

static float g_cos[8][8];
static void fdct_quant(const float* in /* 64 */, const uint16_t* q /* natural */, int16_t* out)
{
    float tmp[64];
    for (int u = 0; u < 8; ++u)
        for (int y = 0; y < 8; ++y) {
            float a = 0;
            for (int x = 0; x < 8; ++x) a += in[y * 8 + x] * g_cos[u][x];
            tmp[y * 8 + u] = a;
        }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) {
            float a = 0;
            for (int y = 0; y < 8; ++y) a += tmp[y * 8 + u] * g_cos[v][y];
            a *= 0.25f * (u ? 1.0f : 0.70710678f) * (v ? 1.0f : 0.70710678f);
            out[v * 8 + u] = (int16_t)lrintf(a / q[v * 8 + u]);
        }
}

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

size_t js_encode(const js_params* p, uint8_t* out, size_t cap)
{
    int hmax = 1, vmax = 1, nc = p->ncomp;
    uint16_t qt[2][64];
    int16_t* coef[4]   = {0, 0, 0, 0};
    int bw[4], bh[4]; /* coefficient array size in blocks: rounded to the interleaved MCU */
    bitw w             = {out, 0, cap, 0, 0, 0};
    htab dc[4], ac[4];
    size_t result = 0;
    if (nc < 1 || nc > 4 || p->width < 1 || p->height < 1 || p->width > 65535 || p->height > 65535) return 0;
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) g_cos[u][x] = cosf((2 * x + 1) * u * 3.14159265358979f / 16);
    for (int c = 0; c < nc; ++c) {
        if (p->hs[c] < 1 || p->hs[c] > 4 || p->vs[c] < 1 || p->vs[c] > 4) return 0;
        if (p->hs[c] > hmax) hmax = p->hs[c];
        if (p->vs[c] > vmax) vmax = p->vs[c];
    }
    {
        int q = p->quality < 1 ? 1 : p->quality > 100 ? 100 : p->quality;
        int scale = q < 50 ? 5000 / q : 200 - 2 * q;
        for (int i = 0; i < 64; ++i) {
            int a = (kLumaQ[i] * scale + 50) / 100, b = (kChromaQ[i] * scale + 50) / 100;
            const int qmax = p->qmax <= 0 ? 127 : p->qmax > 65535 ? 65535 : p->qmax;
            qt[0][i] = (uint16_t)(a < 1 ? 1 : a > qmax ? qmax : a);
            qt[1][i] = (uint16_t)(b < 1 ? 1 : b > qmax ? qmax : b);
        }
    }
    const int single = nc == 1;
    const int mcus_x = ceil_div(p->width, 8 * (single ? 1 : hmax)), mcus_y = ceil_div(p->height, 8 * (single ? 1 : vmax));

    /* paint, subsample, transform */
    for (int c = 0; c < nc; ++c) {
        const int h = single ? 1 : p->hs[c], v = single ? 1 : p->vs[c];
        const int fx = (single ? 1 : hmax) / h, fy = (single ? 1 : vmax) / v; /* box factors (exact for 1,2,4) */
        const int pw = ceil_div(p->width * h, single ? 1 : hmax), phh = ceil_div(p->height * v, single ? 1 : vmax);
        const int interleaved_layout = p->interleaved && !single;
        bw[c] = interleaved_layout ? mcus_x * h : ceil_div(pw, 8);
        bh[c] = interleaved_layout ? mcus_y * v : ceil_div(phh, 8);
        coef[c] = calloc((size_t)bw[c] * bh[c] * 64, sizeof(int16_t));
        float* rows  = malloc(sizeof(float) * (size_t)p->width * (size_t)(fy > 0 ? fy : 1));
        float* strip = malloc(sizeof(float) * (size_t)bw[c] * 8 * 8);
        if (!coef[c] || !rows || !strip) { free(rows); free(strip); goto done; }
        for (int by = 0; by < bh[c]; ++by) {
            for (int r = 0; r < 8; ++r) {
                int sy = by * 8 + r;
                if (sy >= phh) sy = phh - 1; /* replicate the last row (T.81 A.2.4) */
                const int ffy = fy > 0 ? fy : 1, ffx = fx > 0 ? fx : 1;
                for (int k = 0; k < ffy; ++k) {
                    int yy = sy * ffy + k;
                    if (yy >= p->height) yy = p->height - 1;
                    paint_row(p, c, yy, rows + (size_t)k * p->width);
                }
                for (int x = 0; x < bw[c] * 8; ++x) {
                    int sx = x < pw ? x : pw - 1;
                    float a = 0;
                    for (int k = 0; k < ffy; ++k)
                        for (int j = 0; j < ffx; ++j) {
                            int xx = sx * ffx + j;
                            if (xx >= p->width) xx = p->width - 1;
                            a += rows[(size_t)k * p->width + xx];
                        }
                    strip[(size_t)r * bw[c] * 8 + x] = a / (ffx * ffy) - 128.0f;
                }
            }
            for (int bx = 0; bx < bw[c]; ++bx) {
                float blk[64];
                for (int r = 0; r < 8; ++r) memcpy(blk + r * 8, strip + (size_t)r * bw[c] * 8 + bx * 8, 32);
                fdct_quant(blk, qt[c == 0 ? 0 : 1], coef[c] + ((size_t)by * bw[c] + bx) * 64);
            }
        }
        free(rows);
        free(strip);
    }

    /* Huffman tables */
    if (p->optimize) {
        for (int c = 0; c < nc; ++c) {
            long fdc[256], fac[256];
            int pred = 0;
            memset(fdc, 0, sizeof(fdc));
            memset(fac, 0, sizeof(fac));
            /* statistics ignore restart resets of the predictor: tables stay valid, merely sub-optimal */
            for (size_t b = 0; b < (size_t)bw[c] * bh[c]; ++b) code_block(NULL, coef[c] + b * 64, &pred, NULL, NULL, fdc, fac);
            for (int s = 0; s < 12; ++s) if (!fdc[s]) fdc[s] = 1; /* restarts can produce any category */
            fac[0] += 1;
            fac[0xF0] += 1;
            htab_optimal(&dc[c], fdc);
            htab_optimal(&ac[c], fac);
        }
    } else {
        for (int c = 0; c < nc; ++c) {
            if (c == 0) { htab_std(&dc[c], kDcLumaBits, kDcVals, 12); htab_std(&ac[c], kAcLumaBits, kAcLumaVals, 162); }
            else { htab_std(&dc[c], kDcChromaBits, kDcVals, 12); htab_std(&ac[c], kAcChromaBits, kAcChromaVals, 162); }
        }
    }
    /* table ids: optimised -> id = component; standard -> 0 for component 0, 1 otherwise */
    int tid[4];
    for (int c = 0; c < nc; ++c) tid[c] = p->optimize ? c : (c == 0 ? 0 : 1);

    /* headers */
    put_marker(&w, 0xD8);
    {
        int wide = 0;
        for (int t = 0; t < 2; ++t)
            for (int k = 0; k < 64; ++k) wide |= qt[t][k] > 255;
        put_marker(&w, 0xDB);
        put_u16(&w, 2 + 2 * (1 + (wide ? 128 : 64)));
        for (int t = 0; t < 2; ++t) {
            put_byte_raw(&w, (wide ? 0x10 : 0) | t);
            for (int k = 0; k < 64; ++k) {
                if (wide) put_byte_raw(&w, qt[t][kZigzag[k]] >> 8);
                put_byte_raw(&w, qt[t][kZigzag[k]] & 0xFF);
            }
        }
    }
    put_marker(&w, 0xC0);
    put_u16(&w, 8 + 3 * nc);
    put_byte_raw(&w, 8);
    put_u16(&w, p->height);
    put_u16(&w, p->width);
    put_byte_raw(&w, nc);
    for (int c = 0; c < nc; ++c) {
        put_byte_raw(&w, c + 1);
        put_byte_raw(&w, p->hs[c] << 4 | p->vs[c]);
        put_byte_raw(&w, c == 0 ? 0 : 1);
    }
    if (p->restart_interval) {
        put_marker(&w, 0xDD);
        put_u16(&w, 4);
        put_u16(&w, p->restart_interval);
    }

    const int nscans = (p->interleaved || single) ? 1 : nc;
    for (int sidx = 0; sidx < nscans; ++sidx) {
        const int c0 = (p->interleaved || single) ? 0 : sidx, c1 = (p->interleaved || single) ? nc : sidx + 1;
        /* DHT for the tables this scan uses */
        for (int c = c0; c < c1; ++c) {
            int dup = 0;
            for (int d = c0; d < c; ++d) dup |= tid[d] == tid[c];
            if (dup) continue;
            for (int cls = 0; cls < 2; ++cls) {
                const htab* t = cls ? &ac[c] : &dc[c];
                put_marker(&w, 0xC4);
                put_u16(&w, 2 + 1 + 16 + t->count);
                put_byte_raw(&w, cls << 4 | tid[c]);
                for (int i = 0; i < 16; ++i) put_byte_raw(&w, t->bits[i]);
                for (int i = 0; i < t->count; ++i) put_byte_raw(&w, t->vals[i]);
            }
        }
        put_marker(&w, 0xDA);
        put_u16(&w, 6 + 2 * (c1 - c0));
        put_byte_raw(&w, c1 - c0);
        for (int c = c0; c < c1; ++c) {
            put_byte_raw(&w, c + 1);
            put_byte_raw(&w, tid[c] << 4 | tid[c]);
        }
        put_byte_raw(&w, 0);
        put_byte_raw(&w, 63);
        put_byte_raw(&w, 0);

        const int il = (c1 - c0) > 1;
        const int mx = il ? mcus_x : bw[c0], my = il ? mcus_y : bh[c0];
        int pred[4] = {0, 0, 0, 0}, rst = 0, count = 0;
        for (int m = 0; m < mx * my; ++m) {
            if (p->restart_interval && count == p->restart_interval) {
                flush_bits(&w);
                for (int f = 0; f < p->fill_bytes; ++f) put_byte_raw(&w, 0xFF);
                put_marker(&w, 0xD0 + (rst++ & 7));
                memset(pred, 0, sizeof(pred));
                count = 0;
            }
            ++count;
            const int mcx = m % mx, mcy = m / mx;
            for (int c = c0; c < c1; ++c) {
                const int h = il ? p->hs[c] : 1, v = il ? p->vs[c] : 1;
                for (int dy = 0; dy < v; ++dy)
                    for (int dx = 0; dx < h; ++dx) {
                        const size_t b = (size_t)(mcy * v + dy) * bw[c] + (size_t)(mcx * h + dx);
                        code_block(&w, coef[c] + b * 64, &pred[c], &dc[c], &ac[c], NULL, NULL);
                    }
            }
        }
        flush_bits(&w);
    }
    for (int f = 0; f < p->fill_bytes; ++f) put_byte_raw(&w, 0xFF);
    put_marker(&w, 0xD9);
    result = w.overflow ? 0 : w.n;
done:
    for (int c = 0; c < 4; ++c) free(coef[c]);
    return result;
}

/* A grayscale baseline JPEG made of GIVEN quantised coefficient blocks (natural order, DC absolute; DC in
 * [-1024, 1023], AC in [-1023, 1023]) and a given 8-bit quantisation table (natural order), Annex-K luma
 * Huffman tables: blocks_x * blocks_y data units in raster order. Lets a test push chosen coefficients
 * through a decoder's dequantisation + IDCT (tests/golden/idct_kats.npz). */
size_t js_encode_blocks_opt(const int16_t* coef, int blocks_x, int blocks_y, const uint8_t* q, int restart_interval, int optimize, uint8_t* out, size_t cap)
{
    bitw w = {out, 0, cap, 0, 0, 0};
    htab dc, ac;
    if (blocks_x < 1 || blocks_y < 1 || blocks_x > 8191 || blocks_y > 8191) return 0;
    if (optimize) { /* code tables fitted to these very blocks (T.81 K.2): a test can then force the shortest codes */
        long fdc[256] = {0}, fac[256] = {0};
        int pred = 0, count = 0;
        for (int m = 0; m < blocks_x * blocks_y; ++m) {
            if (restart_interval && count == restart_interval) { pred = 0; count = 0; }
            ++count;
            code_block(NULL, coef + (size_t)m * 64, &pred, NULL, NULL, fdc, fac);
        }
        htab_optimal(&dc, fdc);
        htab_optimal(&ac, fac);
    } else {
        htab_std(&dc, kDcLumaBits, kDcVals, 12);
        htab_std(&ac, kAcLumaBits, kAcLumaVals, 162);
    }
    put_marker(&w, 0xD8);
    put_marker(&w, 0xDB);
    put_u16(&w, 2 + 65);
    put_byte_raw(&w, 0);
    for (int k = 0; k < 64; ++k) put_byte_raw(&w, q[kZigzag[k]]);
    put_marker(&w, 0xC0);
    put_u16(&w, 8 + 3);
    put_byte_raw(&w, 8);
    put_u16(&w, blocks_y * 8);
    put_u16(&w, blocks_x * 8);
    put_byte_raw(&w, 1);
    put_byte_raw(&w, 1);
    put_byte_raw(&w, 0x11);
    put_byte_raw(&w, 0);
    if (restart_interval) {
        put_marker(&w, 0xDD);
        put_u16(&w, 4);
        put_u16(&w, restart_interval);
    }
    for (int cls = 0; cls < 2; ++cls) {
        const htab* t = cls ? &ac : &dc;
        put_marker(&w, 0xC4);
        put_u16(&w, 2 + 1 + 16 + t->count);
        put_byte_raw(&w, cls << 4);
        for (int i = 0; i < 16; ++i) put_byte_raw(&w, t->bits[i]);
        for (int i = 0; i < t->count; ++i) put_byte_raw(&w, t->vals[i]);
    }
    put_marker(&w, 0xDA);
    put_u16(&w, 8);
    put_byte_raw(&w, 1);
    put_byte_raw(&w, 1);
    put_byte_raw(&w, 0);
    put_byte_raw(&w, 0);
    put_byte_raw(&w, 63);
    put_byte_raw(&w, 0);
    int pred = 0, rst = 0, count = 0;
    for (int m = 0; m < blocks_x * blocks_y; ++m) {
        if (restart_interval && count == restart_interval) {
            flush_bits(&w);
            put_marker(&w, 0xD0 + (rst++ & 7));
            pred  = 0;
            count = 0;
        }
        ++count;
        code_block(&w, coef + (size_t)m * 64, &pred, &dc, &ac, NULL, NULL);
    }
    flush_bits(&w);
    put_marker(&w, 0xD9);
    return w.overflow ? 0 : w.n;
}

size_t js_encode_blocks(const int16_t* coef, int blocks_x, int blocks_y, const uint8_t* q, int restart_interval, uint8_t* out, size_t cap)
{
    return js_encode_blocks_opt(coef, blocks_x, blocks_y, q, restart_interval, 0, out, cap);
}
