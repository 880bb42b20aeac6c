// fuzz_main.cpp -- mutation fuzzer for the host side of the product (parser, table builder, marker walk,
// work lists) and for the shared symbol loop, driven through the host emulation of the device pipeline.
// Built with -fsanitize=address,undefined by tests/test_fuzz_host.py: any out-of-bounds access, signed
// overflow outside -fwrapv's reach or misaligned read aborts the run. GPU sanitizers are not available
// on the pool, and the device kernels share jg_huff_core.h and the host-built tables with this build.
//
//   fuzz_main <iterations> <seed> file.jpg [file.jpg ...]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int emu_decode_scan(
    const uint8_t* data, size_t size, int subseq_bytes, int max_intra_iters, int scan_idx, int* out_num_subseq,
    int* out_num_du, uint8_t* destuffed, int* seg_index, int* st_p, int* st_n, int* st_cz, int* st_dc, int16_t* coef,
    int* out_max_flow_iters);

static uint64_t g_state = 1;
static uint32_t rnd()
{
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return static_cast<uint32_t>(g_state >> 11);
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const int iterations = std::atoi(argv[1]);
    g_state              = std::strtoull(argv[2], nullptr, 10) * 2654435761u + 88172645463325252ull;
    std::vector<std::vector<uint8_t>> files;
    for (int i = 3; i < argc; ++i) {
        FILE* f = std::fopen(argv[i], "rb");
        if (!f) return 2;
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> d(static_cast<size_t>(n));
        if (std::fread(d.data(), 1, d.size(), f) != d.size()) return 2;
        std::fclose(f);
        files.push_back(d);
    }
    int ok = 0, rejected = 0;
    for (int it = 0; it < iterations; ++it) {
        std::vector<uint8_t> d = files[rnd() % files.size()];
        const int kind         = rnd() % 4;
        const int edits        = 1 + rnd() % 8;
        // header region: everything before the first SOS payload is where the structural damage goes
        size_t hdr = d.size();
        for (size_t i = 0; i + 1 < d.size(); ++i)
            if (d[i] == 0xFF && d[i + 1] == 0xDA) { hdr = i + 16 < d.size() ? i + 16 : d.size(); break; }
        for (int e = 0; e < edits; ++e) {
            const size_t pos = kind == 0 ? rnd() % hdr : kind == 1 ? rnd() % d.size() : hdr + rnd() % (d.size() - hdr + 1);
            if (pos >= d.size()) continue;
            switch (rnd() % 4) {
            case 0: d[pos] = static_cast<uint8_t>(rnd()); break;
            case 1: d[pos] ^= static_cast<uint8_t>(1u << (rnd() % 8)); break;
            case 2: d[pos] = 0xFF; break;
            default: d[pos] = 0; break;
            }
        }
        if (kind == 3 && d.size() > 64) d.resize(d.size() - rnd() % (d.size() / 2)); // truncation
        static const int sizes[3] = {32, 64, 128};
        static const int caps[4]  = {256, 1, 3, 2};
        const int sb = sizes[rnd() % 3], cap = caps[rnd() % 4];
        for (int scan = 0; scan < 4; ++scan) {
            int ns = 0, nd = 0, iters = 0;
            if (emu_decode_scan(d.data(), d.size(), sb, cap, scan, &ns, &nd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) {
                if (scan == 0) ++rejected;
                break;
            }
            if (ns <= 0 || nd <= 0 || static_cast<size_t>(nd) > (size_t{1} << 22)) break; // absurd geometry claims: skip the allocation
            std::vector<uint8_t> dst(static_cast<size_t>(ns) * sb);
            std::vector<int> seg(ns), p(ns), n(ns), cz(ns), dc(4 * static_cast<size_t>(ns));
            std::vector<int16_t> coef(static_cast<size_t>(nd) * 64);
            const int rc = emu_decode_scan(d.data(), d.size(), sb, cap, scan, nullptr, nullptr, dst.data(), seg.data(), p.data(),
                                           n.data(), cz.data(), dc.data(), coef.data(), &iters);
            if (scan == 0) ++(rc == 0 ? ok : rejected);
        }
    }
    std::printf("fuzz: %d iterations, %d decoded, %d rejected\n", iterations, ok, rejected);
    return 0;
}
