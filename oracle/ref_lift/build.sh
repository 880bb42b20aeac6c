#!/bin/bash
# Builds oracle/_ref/libref_idct.so and oracle/_ref/libref_huff.so from the REFERENCE'S OWN SOURCE, where it lies, by line range.
#
# TEST INFRASTRUCTURE (authoring container only: /root/reference does not exist on the GPU box, which
# uses the committed vectors under tests/golden/ and, when it travelled, the prebuilt .so).
#
# The reference as a whole is unbuildable here (nvcc, <cuda_runtime.h>, CUB: DESIGN.md section 5), but the
# arithmetic of its IDCT is a set of self-contained functions -- /root/reference/src/idct.cu:43-144:
# unfixh, unfixo, idct_vector, idct_col, idct_row -- that use nothing but <stdint.h> types. This script
# extracts that range into a temporary directory (nothing of the reference enters the repository or oracle/_ref/),
# and compiles it with g++ and -D__device__= . No stand-in header is written: the only headers involved
# are the C++ standard library's. ref_idct_driver.cpp adds the three statements of the kernel body
# around the lifted functions that cannot be lifted (they sit inside a __global__ function).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${JPEGGPU_REFERENCE:-/root/reference}"
OUT="$HERE/../_ref"
SRC="$REF/src/idct.cu"
[ -f "$SRC" ] || { echo "ref_lift: $SRC not present (not the authoring container): nothing built" >&2; exit 0; }
mkdir -p "$OUT"
TMP="$(mktemp -d)"   # the extracted text lives only for the duration of the build: it is reference source,
trap 'rm -rf "$TMP"' EXIT   # and oracle/_ref/ travels to the GPU box (only the compiled .so may)
sed -n '43,144p' "$SRC" > "$TMP/idct_lifted.inc"
# the range must be what this recipe was written for: five functions, in this order, nothing else
for fn in 'int16_t unfixh(int x)' 'int unfixo(int x)' 'void idct_vector(' 'void idct_col(int16_t\* data, int stride)' 'void idct_row(uint32_t\* v8)'; do
    grep -q "$fn" "$TMP/idct_lifted.inc" || { echo "ref_lift: '$fn' not in $SRC:43-144" >&2; exit 1; }
done
if grep -q '__global__\|<<<\|#include' "$TMP/idct_lifted.inc"; then echo "ref_lift: range is not self-contained" >&2; exit 1; fi
# -fwrapv: device integer arithmetic wraps; -O1 keeps the build quick
g++ -std=c++17 -O1 -fwrapv -fPIC -shared -D__device__= -I"$TMP" -o "$OUT/libref_idct.so" "$HERE/ref_idct_driver.cpp"
echo "$OUT/libref_idct.so"

# ---- the Huffman symbol step, the table builder and the byte rule, the same way -> oracle/_ref/libref_huff.so ----
# defs.hpp:80 (one constant), reader.hpp:45-64 (struct huffman_table), reader.cpp:186-224 (compute_huffman_table),
# decode_huffman.cu:148-286 (u32_select_bits ... decode_next_symbol), decode_destuff.cu:37-44 (is_byte_data). Each range
# is checked for what the recipe expects and for being self-contained; ref_huff_driver.cpp only moves data.
sed -n '80p' "$REF/src/defs.hpp" > "$TMP/defs_lifted.inc"
sed -n '45,64p' "$REF/src/reader.hpp" > "$TMP/table_lifted.inc"
sed -n '186,224p' "$REF/src/reader.cpp" > "$TMP/build_lifted.inc"
sed -n '148,286p' "$REF/src/decode_huffman.cu" > "$TMP/symbol_lifted.inc"
sed -n '37,44p' "$REF/src/decode_destuff.cu" > "$TMP/destuff_lifted.inc"
need() { grep -q "$2" "$TMP/$1" || { echo "ref_lift: '$2' not in the range lifted into $1" >&2; exit 1; }; }
need defs_lifted.inc 'constexpr int huffman_alphabet_size  *= 256;'
need table_lifted.inc '^struct huffman_table {'
need table_lifted.inc 'uint8_t huffval\[huffman_alphabet_size\];'
need build_lifted.inc '^void compute_huffman_table(jpeggpu::huffman_table& table, const uint8_t (&num_codes)\[16\])'
for fn in 'uint32_t u32_select_bits(uint32_t data, int num_bits)' 'uint32_t u32_discard_bits(uint32_t data, int num_bits)' \
          'get_category(uint32_t data, int& length, const huffman_table& table)' 'int get_value(int num_bits, int code)' \
          'void decode_next_symbol_dc(' 'void decode_next_symbol_ac(' 'void decode_next_symbol('; do
    need symbol_lifted.inc "$fn"
done
need destuff_lifted.inc 'bool is_byte_data(bool prev_is_stuffing, uint8_t byte, uint8_t& byte_write)'
[ "$(tail -n 1 "$TMP/table_lifted.inc")" = "};" ] && [ "$(tail -n 1 "$TMP/build_lifted.inc")" = "}" ] && \
    [ "$(tail -n 1 "$TMP/symbol_lifted.inc")" = "}" ] && [ "$(tail -n 1 "$TMP/destuff_lifted.inc")" = "}" ] || { echo "ref_lift: a range does not end where its last definition ends" >&2; exit 1; }
if cat "$TMP/table_lifted.inc" "$TMP/build_lifted.inc" "$TMP/symbol_lifted.inc" "$TMP/destuff_lifted.inc" | grep -q '__global__\|<<<\|#include\|threadIdx'; then
    echo "ref_lift: a range is not self-contained" >&2; exit 1
fi
# C++20: reader.cpp:199-200 uses designated initialisers; asserts stay on (they are part of what is lifted)
g++ -std=c++20 -O1 -fwrapv -fPIC -shared -D__device__= -I"$TMP" -o "$OUT/libref_huff.so" "$HERE/ref_huff_driver.cpp"
echo "$OUT/libref_huff.so"
