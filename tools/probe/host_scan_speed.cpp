#include <emmintrin.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdint>
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(n + 128); if (fread(d.data(), 1, n, f) != (size_t)n) return 1; fclose(f);
    for (int it = 0; it < 5; ++it) {
        auto t0 = std::chrono::steady_clock::now();
        uint64_t cnt = 0, mkc = 0;
        for (long o = 0; o + 65 <= n; o += 64) {
            const uint8_t* blk = d.data() + o;
            uint64_t ff = 0, mk = 0;
            for (int k = 0; k < 4; ++k) {
                const __m128i a  = _mm_loadu_si128(reinterpret_cast<const __m128i*>(blk + 16 * k));
                const __m128i nn  = _mm_loadu_si128(reinterpret_cast<const __m128i*>(blk + 16 * k + 1));
                const __m128i fm  = _mm_cmpeq_epi8(a, _mm_set1_epi8(static_cast<char>(0xFF)));
                const __m128i nz = _mm_cmpeq_epi8(nn, _mm_setzero_si128());
                ff |= static_cast<uint64_t>(static_cast<uint32_t>(_mm_movemask_epi8(fm))) << (16 * k);
                mk |= static_cast<uint64_t>(static_cast<uint32_t>(_mm_movemask_epi8(_mm_andnot_si128(nz, fm)))) << (16 * k);
            }
            cnt += __builtin_popcountll(ff); mkc += __builtin_popcountll(mk);
        }
        auto t1 = std::chrono::steady_clock::now();
        // memchr walk
        uint64_t c2 = 0; const uint8_t* p = d.data(); const uint8_t* e = d.data() + n;
        while (true) { const uint8_t* q = (const uint8_t*)memchr(p, 0xFF, e - p); if (!q) break; ++c2; p = q + 1; }
        auto t2 = std::chrono::steady_clock::now();
        printf("sse scan %.3f ms (ff %lu mk %lu)   memchr walk %.3f ms (%lu)\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), cnt, mkc,
            std::chrono::duration<double, std::milli>(t2 - t1).count(), c2);
    }
}
