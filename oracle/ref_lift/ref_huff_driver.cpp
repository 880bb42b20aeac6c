// ref_huff_driver.cpp -- C entry points around the reference's own Huffman symbol step, table builder and byte rule.
//
// TEST INFRASTRUCTURE. The five .inc files are line ranges of /root/reference/src, extracted by build.sh at build time
// into a temporary directory (never committed, never shipped), compiled as they stand with -D__device__= :
//   defs_lifted.inc     defs.hpp:80            constexpr int huffman_alphabet_size = 256;
//   table_lifted.inc    reader.hpp:45-64       struct huffman_table
//   build_lifted.inc    reader.cpp:186-224     compute_huffman_table
//   symbol_lifted.inc   decode_huffman.cu:148-286  u32_select_bits, u32_discard_bits, get_category, get_value,
//                                               decode_next_symbol_dc / _ac, decode_next_symbol
//   destuff_lifted.inc  decode_destuff.cu:37-44    is_byte_data
// Everything that decides a result below is THEIR code; this file only moves data in and out.
#include <stdint.h>

#include <cassert>
#include <cstring>

namespace jpeggpu {
#include "defs_lifted.inc"
#include "table_lifted.inc"
} // namespace jpeggpu
using namespace jpeggpu;
#include "build_lifted.inc"
namespace {
#include "symbol_lifted.inc"
#include "destuff_lifted.inc"
} // namespace

extern "C" {

int ref_huff_table_bytes() { return static_cast<int>(sizeof(huffman_table)); }

/// compute_huffman_table on a DHT payload: counts per code length, then the values in code order (`count` of them; the
/// rest of huffval is zeroed as reader.cpp:295-297 does). `table` receives the reference's struct, byte for byte.
void ref_huff_build(const uint8_t* num_codes, const uint8_t* huffval, int count, void* table)
{
    huffman_table t;
    std::memset(&t, 0, sizeof(t));
    for (int i = 0; i < count && i < 256; ++i) t.huffval[i] = huffval[i];
    uint8_t nc[16];
    std::memcpy(nc, num_codes, 16);
    compute_huffman_table(t, nc);
    std::memcpy(table, &t, sizeof(t));
}

/// decode_next_symbol<true> on n windows (decode_huffman.cu:273-286: z == 0 selects the DC table). Both table arguments
/// are `table`: only the one the index selects is read.
void ref_huff_symbols(const void* table, const uint32_t* data, const int* z, int n, int* length, int* symbol, int* run_length)
{
    huffman_table t;
    std::memcpy(&t, table, sizeof(t));
    for (int i = 0; i < n; ++i) {
        int l = 0, s = 0, r = 0;
        decode_next_symbol<true>(l, s, r, data[i], t, t, z[i]);
        length[i]     = l;
        symbol[i]     = s;
        run_length[i] = r;
    }
}

/// is_byte_data for n (previous byte, byte) pairs; prev_is_stuffing as its callers form it (decode_destuff.cu:64, :91:
/// `scan_stuffed[tid - 1] == 0xff`).
void ref_byte_rule(const uint8_t* prev, const uint8_t* byte, int n, uint8_t* is_data, uint8_t* written)
{
    for (int i = 0; i < n; ++i) {
        uint8_t w   = 0;
        const bool d = is_byte_data(prev[i] == 0xff, byte[i], w);
        is_data[i]  = d ? 1 : 0;
        written[i]  = d ? w : 0; // (what it would write for a byte that is not data is never stored)
    }
}

} // extern "C"
