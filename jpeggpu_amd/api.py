"""ctypes mirror of include/jpeggpu/jpeggpu.h (+ jpeggpu_ext.h).

Call order is the reference's (example/example_tool.c:101-128):
    startup -> parse_header -> get_buffer_size -> transfer(d_tmp) -> decode(d_tmp) -> cleanup.
Device memory and streams are the caller's: pass raw device pointers (e.g. torch tensors'
data_ptr()) and a hipStream_t handle (torch.cuda.Stream.cuda_stream). There is no CPU fallback: if
the HIP library is missing, importing the binding fails.
"""
import ctypes as C
import enum
import os

from . import build as _build

MAX_COMP = 4
STAGES = ("front", "destuff", "sync_intra", "sync_inter", "tails", "write", "idct")


class Status(enum.IntEnum):
    SUCCESS = 0
    INVALID_ARGUMENT = 1
    INVALID_JPEG = 2
    INTERNAL_ERROR = 3
    NOT_SUPPORTED = 4
    OUT_OF_HOST_MEMORY = 5
    INCOMPLETE_BITSTREAM = 6


class Subsampling(C.Structure):
    _fields_ = [("x", C.c_int * MAX_COMP), ("y", C.c_int * MAX_COMP)]


class ImgInfo(C.Structure):
    _fields_ = [("sizes_x", C.c_int * MAX_COMP), ("sizes_y", C.c_int * MAX_COMP),
                ("num_components", C.c_int), ("subsampling", Subsampling)]


class Img(C.Structure):
    _fields_ = [("image", C.c_void_p * MAX_COMP), ("pitch", C.c_int * MAX_COMP)]


class ExtScanLayout(C.Structure):
    _fields_ = [
        ("num_components", C.c_int), ("component_idx", C.c_int * MAX_COMP),
        ("num_subsequences", C.c_int), ("num_segments", C.c_int), ("num_sequences", C.c_int),
        ("num_data_units", C.c_int), ("data_units_per_mcu", C.c_int), ("num_chunks", C.c_int),
        ("off_segments", C.c_size_t), ("off_chunks", C.c_size_t), ("off_destuffed", C.c_size_t),
        ("off_segment_index", C.c_size_t), ("off_state_p", C.c_size_t), ("off_state_n", C.c_size_t),
        ("off_state_cz", C.c_size_t), ("off_state_dc01", C.c_size_t), ("off_state_dc23", C.c_size_t),
        ("off_symbols", C.c_size_t), ("off_du_table", C.c_size_t), ("symbol_region_entries", C.c_int),
        ("device_scan", C.c_int), ("off_device_status", C.c_size_t),
        ("hypotheses", C.c_int), ("hypothesis_blocks", C.c_int),
    ]


class ExtLayout(C.Structure):
    _fields_ = [
        ("subsequence_bytes", C.c_int), ("subsequences_per_sequence", C.c_int), ("num_scans", C.c_int),
        ("transferred_bytes", C.c_size_t), ("blob_bytes", C.c_size_t),
        ("off_bytes", C.c_size_t), ("off_qtables", C.c_size_t),
        ("scans", ExtScanLayout * MAX_COMP),
        ("shard_rank", C.c_int), ("shard_world", C.c_int),
    ]


class ParseItem(C.Structure):
    _fields_ = [("decoder", C.c_void_p), ("img_info", C.POINTER(ImgInfo)), ("data", C.c_void_p), ("size", C.c_size_t)]


class BatchItem(C.Structure):
    _fields_ = [("decoder", C.c_void_p), ("img", C.POINTER(Img)), ("d_tmp", C.c_void_p), ("tmp_size", C.c_size_t)]


class JpegGpuError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = Status(status)
        super().__init__("%s: %s" % (where, status_string(status)))


_lib = None


def lib():
    """Load the C-ABI library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64 (SONAME libamdhip64.so.7, loaded via RPATH under the file name
    # libamdhip64.so). Import it first so our NEEDED libamdhip64.so.7 resolves to that same runtime:
    # two HIP runtimes in one process do not know each other's allocations and streams.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("JPEGGPU_LIB", _build.LIB_PATH)  # override: A/B runs of experimental builds
    if not os.path.exists(path):
        raise ImportError(
            "jpeggpu_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    L = C.CDLL(path)
    dec = C.c_void_p
    L.jpeggpu_get_status_string.restype = C.c_char_p
    L.jpeggpu_get_status_string.argtypes = [C.c_int]
    L.jpeggpu_decoder_startup.argtypes = [C.POINTER(dec)]
    L.jpeggpu_set_logging.argtypes = [dec, C.c_int]
    L.is_css_444.argtypes = [Subsampling, C.c_int]
    L.jpeggpu_decoder_parse_header.argtypes = [dec, C.POINTER(ImgInfo), C.c_void_p, C.c_size_t]
    L.jpeggpu_decoder_get_buffer_size.argtypes = [dec, C.POINTER(C.c_size_t)]
    L.jpeggpu_decoder_transfer.argtypes = [dec, C.c_void_p, C.c_size_t, C.c_void_p]
    L.jpeggpu_decoder_decode.argtypes = [dec, C.POINTER(Img), C.c_void_p, C.c_size_t, C.c_void_p]
    L.jpeggpu_decoder_cleanup.argtypes = [dec]
    L.jpeggpu_ext_set_subsequence_bytes.argtypes = [dec, C.c_int]
    L.jpeggpu_ext_set_batched.argtypes = [dec, C.c_int]
    if hasattr(L, "jpeggpu_ext_set_batch_hint"):  # (a library built before round 5, loaded through JPEGGPU_LIB for an A/B run, has none)
        L.jpeggpu_ext_set_batch_hint.argtypes = [dec, C.c_int]
    L.jpeggpu_ext_get_layout.argtypes = [dec, C.POINTER(ExtLayout)]
    L.jpeggpu_ext_set_profiling.argtypes = [dec, C.c_int]
    L.jpeggpu_ext_get_stage_ms.argtypes = [dec, C.POINTER(C.c_float)]
    L.jpeggpu_ext_batch_scratch_size.restype = C.c_size_t
    L.jpeggpu_ext_batch_scratch_size.argtypes = [C.c_int]
    L.jpeggpu_ext_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.jpeggpu_ext_decode_batch.argtypes = [C.c_void_p, C.POINTER(BatchItem), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.jpeggpu_ext_batch_destroy.argtypes = [C.c_void_p]
    L.jpeggpu_ext_batch_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.jpeggpu_ext_batch_set_sync_iterations.argtypes = [C.c_void_p, C.c_int]
    L.jpeggpu_ext_batch_set_overlap.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "jpeggpu_ext_batch_set_fused_tail"):  # (experimental builds of older trees, JPEGGPU_LIB, lack it)
        L.jpeggpu_ext_batch_set_fused_tail.argtypes = [C.c_void_p, C.c_int]
        L.jpeggpu_ext_fused_tail_timeouts.argtypes = [C.POINTER(C.c_uint)]
    L.jpeggpu_ext_batch_get_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.jpeggpu_ext_upsample_planes.argtypes = [
        C.POINTER(ImgInfo), C.POINTER(Img), C.POINTER(Img), C.c_int, C.c_int, C.c_void_p]
    L.jpeggpu_ext_set_segment_shard.argtypes = [dec, C.c_int, C.c_int]
    L.jpeggpu_ext_get_shard_rows.argtypes = [dec, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.jpeggpu_ext_self_test.argtypes = [C.c_void_p]
    L.jpeggpu_ext_set_device_scan.argtypes = [dec, C.c_int]
    L.jpeggpu_ext_get_device_status.argtypes = [dec, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.jpeggpu_ext_parse_headers.argtypes = [C.POINTER(ParseItem), C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.jpeggpu_ext_planes_to_rgbi.argtypes = [
        C.POINTER(ImgInfo), C.POINTER(Img), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _lib = L
    return L


def status_string(status) -> str:
    return lib().jpeggpu_get_status_string(int(status)).decode()


def _check(status, where):
    if status != 0:
        raise JpegGpuError(status, where)


def _host_buffer(data):
    """(address, size, object to keep alive) of a bytes / bytearray input. The decoder borrows the address until
    the copy enqueued by transfer() has executed, so the address must belong to the object that is kept: the
    bytes object itself, or a ctypes view of the bytearray's own storage (no temporary copy)."""
    if isinstance(data, bytearray):
        view = (C.c_char * len(data)).from_buffer(data) if len(data) else (C.c_char * 1)()
        return C.addressof(view), len(data), (data, view)
    return C.cast(C.c_char_p(data), C.c_void_p).value, len(data), data


class Decoder:
    """One jpeggpu_decoder_t. Not thread-safe; one decoder per host thread / stream."""

    def __init__(self, subseq_bytes=None):
        self._h = C.c_void_p()
        _check(lib().jpeggpu_decoder_startup(C.byref(self._h)), "jpeggpu_decoder_startup")
        self._keep = None
        if subseq_bytes is not None:
            self.set_subsequence_bytes(subseq_bytes)

    def set_logging(self, on: bool):
        _check(lib().jpeggpu_set_logging(self._h, int(on)), "jpeggpu_set_logging")

    def set_subsequence_bytes(self, n: int):
        _check(lib().jpeggpu_ext_set_subsequence_bytes(self._h, n), "jpeggpu_ext_set_subsequence_bytes")

    def set_batched(self, on: bool = True):
        """This decoder's images share their launches with others (jpeggpu_ext_decode_batch): the per-image choice
        of the subsequence size is made for throughput instead of for the latency of one image."""
        _check(lib().jpeggpu_ext_set_batched(self._h, int(on)), "jpeggpu_ext_set_batched")

    def set_batch_hint(self, images_per_call: int):
        """About how many images of this kind share one jpeggpu_ext_decode_batch call (0: decoded on its own). The plan
        of the next parsed images -- subsequence size, multi-hypothesis speculation -- is made for a launch of that size."""
        if hasattr(lib(), "jpeggpu_ext_set_batch_hint"):
            _check(lib().jpeggpu_ext_set_batch_hint(self._h, int(images_per_call)), "jpeggpu_ext_set_batch_hint")
        else:
            self.set_batched(images_per_call > 0)

    def parse_header(self, data, size=None) -> ImgInfo:
        """`data`: bytes, a numpy uint8 array, or an integer host address (then `size` is required).
        The buffer is borrowed until the copy enqueued by transfer() has executed."""
        info = ImgInfo()
        if isinstance(data, int):
            ptr, n = data, size
        elif isinstance(data, (bytes, bytearray)):
            ptr, n, self._keep = _host_buffer(data)
        else:  # numpy / torch-like with ctypes or data_ptr
            self._keep = data
            ptr = data.ctypes.data if hasattr(data, "ctypes") else data.data_ptr()
            n = data.nbytes if hasattr(data, "nbytes") else data.numel()
        _check(lib().jpeggpu_decoder_parse_header(self._h, C.byref(info), ptr, n), "jpeggpu_decoder_parse_header")
        return info

    def set_device_scan(self, on=True):
        """False / 0: host walk; True / 1: marker scan on the device, status via device_status(); 2: checked --
        decode() waits for the stream and raises the device's status (what JPEGGPU_DEVICE_SCAN=2 selects)."""
        _check(lib().jpeggpu_ext_set_device_scan(self._h, int(on)), "jpeggpu_ext_set_device_scan")

    def set_segment_shard(self, rank: int, world: int):
        """Decode only restart segments [rank * n / world, (rank + 1) * n / world) of the next parsed images."""
        _check(lib().jpeggpu_ext_set_segment_shard(self._h, rank, world), "jpeggpu_ext_set_segment_shard")

    def shard_rows(self, component: int):
        """(first_row, num_rows) of the plane rows this decoder writes."""
        a, n = C.c_int(), C.c_int()
        _check(lib().jpeggpu_ext_get_shard_rows(self._h, component, C.byref(a), C.byref(n)), "jpeggpu_ext_get_shard_rows")
        return a.value, n.value

    def device_status(self, d_tmp: int, stream: int = 0) -> Status:
        st = C.c_int()
        _check(lib().jpeggpu_ext_get_device_status(self._h, d_tmp, stream, C.byref(st)), "jpeggpu_ext_get_device_status")
        return Status(st.value)

    def get_buffer_size(self) -> int:
        n = C.c_size_t()
        _check(lib().jpeggpu_decoder_get_buffer_size(self._h, C.byref(n)), "jpeggpu_decoder_get_buffer_size")
        return n.value

    def transfer(self, d_tmp: int, tmp_size: int, stream: int = 0):
        _check(lib().jpeggpu_decoder_transfer(self._h, d_tmp, tmp_size, stream), "jpeggpu_decoder_transfer")

    def decode(self, planes, pitches, d_tmp: int, tmp_size: int, stream: int = 0):
        img = Img()
        for c, (p, pitch) in enumerate(zip(planes, pitches)):
            img.image[c] = p
            img.pitch[c] = pitch
        _check(lib().jpeggpu_decoder_decode(self._h, C.byref(img), d_tmp, tmp_size, stream), "jpeggpu_decoder_decode")

    def set_profiling(self, on: bool):
        _check(lib().jpeggpu_ext_set_profiling(self._h, int(on)), "jpeggpu_ext_set_profiling")

    def stage_ms(self):
        """Per-stage milliseconds of the last decode (after the stream was synchronised)."""
        ms = (C.c_float * len(STAGES))()
        _check(lib().jpeggpu_ext_get_stage_ms(self._h, ms), "jpeggpu_ext_get_stage_ms")
        return dict(zip(STAGES, ms))

    def layout(self) -> ExtLayout:
        lay = ExtLayout()
        _check(lib().jpeggpu_ext_get_layout(self._h, C.byref(lay)), "jpeggpu_ext_get_layout")
        return lay

    def cleanup(self):
        if self._h:
            lib().jpeggpu_decoder_cleanup(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


class Batch:
    """jpeggpu_ext_decode_batch: one launch per stage for many parsed + transferred images."""

    def __init__(self, max_scans: int):
        self._h = C.c_void_p()
        self.max_scans = max_scans
        _check(lib().jpeggpu_ext_batch_create(C.byref(self._h), max_scans), "jpeggpu_ext_batch_create")
        self.scratch_size = lib().jpeggpu_ext_batch_scratch_size(max_scans)
        self._items = None
        self._keep = None

    def set_items(self, entries):
        """entries: list of (Decoder, plane_ptrs, pitches, d_tmp, tmp_size). Built once, reused per call."""
        n = len(entries)
        arr = (BatchItem * n)()
        imgs = []
        for i, (dec, ptrs, pitches, d_tmp, tmp_size) in enumerate(entries):
            img = Img()
            for c, (p, pitch) in enumerate(zip(ptrs, pitches)):
                img.image[c] = p
                img.pitch[c] = pitch
            imgs.append(img)
            arr[i].decoder = dec._h
            arr[i].img = C.pointer(img)
            arr[i].d_tmp = d_tmp
            arr[i].tmp_size = tmp_size
        self._items, self._keep = arr, (imgs, entries)

    def decode(self, d_scratch: int, stream: int = 0):
        _check(lib().jpeggpu_ext_decode_batch(self._h, self._items, len(self._items), d_scratch, self.scratch_size, stream),
               "jpeggpu_ext_decode_batch")

    def set_overlap(self, parts: int):
        _check(lib().jpeggpu_ext_batch_set_overlap(self._h, parts), "jpeggpu_ext_batch_set_overlap")

    def set_fused_tail(self, enable: bool):
        _check(lib().jpeggpu_ext_batch_set_fused_tail(self._h, 1 if enable else 0), "jpeggpu_ext_batch_set_fused_tail")

    def set_sync_iterations(self, n: int):
        _check(lib().jpeggpu_ext_batch_set_sync_iterations(self._h, n), "jpeggpu_ext_batch_set_sync_iterations")

    def set_profiling(self, on: bool):
        _check(lib().jpeggpu_ext_batch_set_profiling(self._h, int(on)), "jpeggpu_ext_batch_set_profiling")

    def stage_ms(self):
        ms = (C.c_float * len(STAGES))()
        _check(lib().jpeggpu_ext_batch_get_stage_ms(self._h, ms), "jpeggpu_ext_batch_get_stage_ms")
        return dict(zip(STAGES, ms))

    def destroy(self):
        if self._h:
            lib().jpeggpu_ext_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def fused_tail_timeouts() -> int:
    """Writers of the fused tail + write launch that gave up waiting, since the library was loaded (0 on a correct run)."""
    n = C.c_uint(0)
    _check(lib().jpeggpu_ext_fused_tail_timeouts(C.byref(n)), "jpeggpu_ext_fused_tail_timeouts")
    return n.value


def self_test(stream: int = 0) -> None:
    """jpeggpu_ext_self_test: one small built-in decode checked against stored plane hashes; raises JpegGpuError if the
    running system does not decode bit-exactly."""
    _check(lib().jpeggpu_ext_self_test(stream), "jpeggpu_ext_self_test")


def decode_to_planes(data: bytes, device="cuda:0", subseq_bytes=None, return_tmp=False, device_scan=False):
    """Convenience wrapper used by tests: full call sequence on torch's current stream, returns the
    planes as torch uint8 tensors on `device` (torch is only the allocator / stream provider). With
    `device_scan` the restart markers are found on the device and a status it reports there is raised."""
    import torch

    dec = Decoder(subseq_bytes)
    try:
        if device_scan:
            dec.set_device_scan(True)
        info = dec.parse_header(data)
        n = dec.get_buffer_size()
        tmp = torch.empty(n + 256, dtype=torch.uint8, device=device)
        base = (tmp.data_ptr() + 255) // 256 * 256
        stream = torch.cuda.current_stream(torch.device(device)).cuda_stream
        planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device=device)
                  for c in range(info.num_components)]
        dec.transfer(base, n, stream)
        dec.decode([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, stream)
        torch.cuda.synchronize(torch.device(device))
        if device_scan:
            _check(int(dec.device_status(base, stream)), "device-side marker scan")
        if return_tmp:
            return planes, info, tmp, base, dec.layout()
        return planes, info
    finally:
        dec.cleanup()


def parse_headers(decoders, buffers, num_threads=4):
    """jpeggpu_ext_parse_headers: parse `buffers[i]` (bytes or numpy uint8; kept alive by the decoder object)
    into `decoders[i]` on a pool of host threads. Returns the list of ImgInfo; raises on the first failure."""
    n = len(decoders)
    items = (ParseItem * n)()
    infos = [ImgInfo() for _ in range(n)]
    for i, (d, b) in enumerate(zip(decoders, buffers)):
        d._keep = b
        if isinstance(b, (bytes, bytearray)):
            ptr, size, d._keep = _host_buffer(b)
        else:
            ptr = b.ctypes.data if hasattr(b, "ctypes") else b.data_ptr()
            size = b.nbytes if hasattr(b, "nbytes") else b.numel()
        items[i].decoder, items[i].img_info, items[i].data, items[i].size = d._h.value, C.pointer(infos[i]), ptr, size
    st = (C.c_int * n)()
    _check(lib().jpeggpu_ext_parse_headers(items, n, num_threads, st), "jpeggpu_ext_parse_headers")
    return infos
