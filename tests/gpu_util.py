"""Helpers of the GPU tests: reading intermediate buffers of a decode back through jpeggpu_ext_get_layout."""
import numpy as np

NATURAL = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                    6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45,
                    38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


def tmp_view(torch, tmp, base, off, count, dtype):
    start = base - tmp.data_ptr() + off
    nbytes = count * torch.tensor([], dtype=dtype).element_size()
    return tmp[start:start + nbytes].view(dtype).cpu().numpy()


def stream_coefficients(torch, tmp, base, sl, num_subseq=None):
    """Dense stream-order coefficients int16 [num_data_units, 64] (natural order, DC absolute) rebuilt from what
    the write pass leaves in d_tmp: the symbol stream (16-bit entries; regions of 64 subsequences interleaved in
    sectors of 16 entries, jpeggpu_ext.h) and the data-unit table {physical index of the first entry, count | 128 if
    the unit holds an escape}. A unit's first entry is its DC value; an AC entry is value << 6 | zig-zag index (10-bit
    value); an entry with index 0 behind one is an escape carrying value >> 10 in its high bits."""
    S = sl.num_subsequences if num_subseq is None else num_subseq
    ND = sl.num_data_units
    nsym = ((S + 63) // 64) * (sl.symbol_region_entries // 16) * 1024
    sym = tmp_view(torch, tmp, base, sl.off_symbols, nsym, torch.int16).view(np.uint16).astype(np.uint32)
    tab = tmp_view(torch, tmp, base, sl.off_du_table, ND * 2, torch.int32).view(np.uint32).reshape(ND, 2)
    assert tab[:, 1].max() <= 255 and (tab[:, 1] & 127).min() >= 1
    coef = np.zeros((ND, 64), np.int16)
    cnt = (tab[:, 1] & 127).astype(np.int64)
    du = np.repeat(np.arange(ND), cnt)
    k = np.arange(len(du), dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)  # index of the entry in its unit
    first = np.repeat(tab[:, 0].astype(np.int64), cnt)
    w = (first & 15) + k
    idx = (first & ~15) + (w >> 4) * 1024 + (w & 15)
    assert idx.max() < sym.size
    ent = sym[idx]
    zz = ent & 63
    is_dc = k == 0
    is_esc = (~is_dc) & (zz == 0)
    nxt_esc = np.zeros(len(ent), bool)
    nxt_esc[:-1] = is_esc[1:] & (du[1:] == du[:-1])
    nxt = np.zeros(len(ent), np.uint32)
    nxt[:-1] = ent[1:]
    ent = ent.astype(np.int64)
    nxt = nxt.astype(np.int64)
    zz = zz.astype(np.int64)
    val = np.where(nxt_esc, (((nxt >> 6) << 10) | (ent >> 6)) & 0xFFFF, ((ent >> 6) ^ 0x200) - 0x200 + 0x10000) & 0xFFFF
    val = np.where(is_dc, ent, val).astype(np.uint16).view(np.int16)
    keep = ~is_esc
    coef[du[keep], NATURAL[np.where(is_dc, 0, zz)[keep]]] = val[keep]
    has_esc = np.zeros(ND, bool)
    has_esc[du[is_esc]] = True
    assert np.array_equal(has_esc, (tab[:, 1] & 128) != 0), "escape flag of the data-unit records"
    return coef


def component_blocks(info, lay, scan_idx, stream_coef):
    """Stream order -> {component index: int16 [blocks_y, blocks_x, 64]} for one scan (T.81 A.2.3: an interleaved
    scan codes MCU after MCU, components in scan order, v x h data units row-major each; a scan of one component
    codes its ceil(size / 8) blocks in raster order)."""
    sl = lay.scans[scan_idx]
    nc = info.num_components
    comps = [sl.component_idx[k] for k in range(sl.num_components)]
    out = {}
    if len(comps) == 1:
        c = comps[0]
        bx, by = (info.sizes_x[c] + 7) // 8, (info.sizes_y[c] + 7) // 8
        assert bx * by == len(stream_coef)
        out[c] = stream_coef.reshape(by, bx, 64)
        return out
    hmax = max(info.subsampling.x[c] for c in range(nc))
    vmax = max(info.subsampling.y[c] for c in range(nc))
    # img_info has no frame size; the plane of a component with the maximum factor has exactly the frame's extent
    c0 = max(range(nc), key=lambda c: info.subsampling.x[c])
    width = info.sizes_x[c0] * hmax // info.subsampling.x[c0]
    c1 = max(range(nc), key=lambda c: info.subsampling.y[c])
    height = info.sizes_y[c1] * vmax // info.subsampling.y[c1]
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    dpm = sl.data_units_per_mcu
    assert mx * my * dpm == len(stream_coef), (mx, my, dpm, len(stream_coef))
    s = stream_coef.reshape(my, mx, dpm, 64)
    k = 0
    for c in comps:
        h, v = info.subsampling.x[c], info.subsampling.y[c]
        blk = s[:, :, k:k + h * v].reshape(my, mx, v, h, 64).transpose(0, 2, 1, 3, 4).reshape(my * v, mx * h, 64)
        out[c] = blk
        k += h * v
    return out
