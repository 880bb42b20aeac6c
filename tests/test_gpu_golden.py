"""The committed golden fixtures (tests/golden/pin_vectors.npz: 17 JPEGs from three encoders -- this
repository's, Pillow / libjpeg-turbo, IJG cjpeg -- with IJG libjpeg 9d's jpeg_read_coefficients output) through
the HIP path. Foreign-encoder coverage as the reference's test/test.sh:31-43 (ImageMagick variants) has it.

Two independent checks per file:
  * the quantised coefficients the HIP Huffman + DC path produced, read back from the symbol stream through
    jpeggpu_ext_get_layout, == the committed IJG arrays -- the oracle is not involved;
  * planes == oracle (bit-exact), the oracle's IDCT itself being pinned by tests/test_idct_kats.py.
"""
import os

import numpy as np
import pytest

from tests import gpu_util
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vectors():
    z = np.load(os.path.join(GOLDEN, "pin_vectors.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    assert "cjpeg_ni_420" in names and "pil_cmyk" in names and len(names) >= 17
    return z, names


@pytest.mark.parametrize("subseq_bytes,device_scan", [(128, False), (64, True), (32, False)])
def test_golden_fixtures_coefficients_and_planes(gpu_lib, vectors, subseq_bytes, device_scan):
    import torch

    import jpeggpu_amd
    from oracle import oracle

    z, names = vectors
    bad = []
    for n in names:
        data = z[n + "/jpeg"].tobytes()
        planes, info, tmp, base, lay = jpeggpu_amd.decode_to_planes(
            data, subseq_bytes=subseq_bytes, return_tmp=True, device_scan=device_scan)
        seen = set()
        for s in range(lay.num_scans):
            sl = lay.scans[s]
            S = None
            if sl.device_scan:  # the layout holds capacities, the device reports what it found
                words = gpu_util.tmp_view(torch, tmp, base, sl.off_device_status, 5, torch.int32)
                assert words[0] == 0
                S = int(words[1])
            coef = gpu_util.stream_coefficients(torch, tmp, base, sl, S)
            for c, blk in gpu_util.component_blocks(info, lay, s, coef).items():
                want = z["%s/coef%d" % (n, c)]
                if not np.array_equal(blk[:want.shape[0], :want.shape[1]], want):
                    bad.append((n, "coef", c))
                seen.add(c)
        assert seen == set(range(info.num_components)), n
        ref = oracle.decode(data)
        for c in range(ref.ncomp):
            if not np.array_equal(planes[c].cpu().numpy(), ref.planes[c]):
                bad.append((n, "plane", c))
    assert not bad, bad
