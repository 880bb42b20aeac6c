#!/bin/bash
# Builds oracle/_ref/libref_idct.so from the REFERENCE'S OWN SOURCE, where it lies, by line range.
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
