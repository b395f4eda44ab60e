"""Host-side mirror of the reference's Go API for the batch hot path.

Every function cites the Go source it mirrors.  All data movement goes through libzstd_amd.so
(GPU); nothing here compresses or decompresses on the CPU.
"""
import ctypes

from ._lib import lib, c_size_p, c_void_pp

# zstd.go:20-24
BestSpeed = 1
BestCompression = 20
DefaultCompression = 5

# zstd.go:31-40
_decompressSizeBufferLimit = 1000 * 1000
_zstdFrameHeaderSizeMin = 2


class ZstdError(Exception):
    pass


class ErrorCode(ZstdError):
    """errors.go:9-19 -- an error code returned by the C library; str() is ZSTD_getErrorName."""

    def __init__(self, code):
        self.code = int(code)
        super().__init__(lib.zsa_getErrorName(ctypes.c_size_t(self.code & (2 ** 64 - 1))).decode())

    def __eq__(self, other):
        return isinstance(other, ErrorCode) and other.code == self.code

    def __hash__(self):
        return hash(self.code)


ErrEmptySlice = ZstdError("Bytes slice is empty")            # zstd.go:28
ErrEmptyDictionary = ZstdError("Dictionary is empty")        # zstd_bulk.go:15
ErrBadDictionary = ZstdError("Cannot load dictionary")       # zstd_bulk.go:17


def _get_error(code):
    """errors.go:21-26 getError: code<0 as a signed int && ZSTD_isError."""
    if lib.zsa_isError(code):
        return ErrorCode(code - 2 ** 64)
    return None


def IsDstSizeTooSmallError(e):
    """errors.go:29-34: string match on the error name."""
    return isinstance(e, ErrorCode) and str(e) == "Destination buffer is too small"


def CompressBound(srcSize):
    """zstd.go:46-53 (Go re-implementation of ZSTD_COMPRESSBOUND)."""
    lowLimit = 128 << 10
    margin = (lowLimit - srcSize) >> 11 if srcSize < lowLimit else 0
    return srcSize + (srcSize >> 8) + margin


def cCompressBound(srcSize):
    """zstd.go:56-58."""
    return lib.zsa_compressBound(srcSize)


def _buf(b):
    """bytes-like -> (ctypes pointer, keepalive)."""
    if isinstance(b, (bytes, bytearray, memoryview)):
        n = len(b)
        arr = (ctypes.c_char * max(n, 1)).from_buffer_copy(bytes(b) if n else b"\0")
        return ctypes.cast(arr, ctypes.c_void_p), arr
    raise TypeError("expected bytes-like")


def decompressSizeHint(src):
    """zstd.go:62-85."""
    upperBound = max(10 * len(src), _decompressSizeBufferLimit)
    hint = upperBound
    if len(src) >= _zstdFrameHeaderSizeMin:
        p, keep = _buf(src)
        h = lib.zsa_getFrameContentSize(p, len(src))
        hint = h - 2 ** 64 if h >= 2 ** 63 else h      # Go: int(C.ZSTD_getFrameContentSize(...))
        if hint < 0:
            hint = upperBound
        if hint == 0:
            hint = 1
    return min(hint, upperBound)


def CompressLevel(dst, src, level):
    """zstd.go:95-129.  `dst` (bytearray or None) is reused when its length >= CompressBound(len(src))."""
    bound = CompressBound(len(src))
    out = (ctypes.c_char * max(bound, 1))()
    sp, keep = _buf(src)
    written = lib.zsa_compress(ctypes.cast(out, ctypes.c_void_p), bound, sp if len(src) else None, len(src), level)
    err = _get_error(written)
    if err is not None:
        raise err
    res = out.raw[:written]
    if dst is not None and isinstance(dst, bytearray) and len(dst) >= bound:
        dst[:written] = res
        return memoryview(dst)[:written]
    return res


def Compress(dst, src):
    """zstd.go:90-92."""
    return CompressLevel(dst, src, DefaultCompression)


def DecompressInto(dst, src):
    """zstd.go:167-174: dst is a bytearray of the wanted capacity; returns bytes written."""
    cap = len(dst)
    out = (ctypes.c_char * max(cap, 1))()
    sp, keep = _buf(src)
    written = lib.zsa_decompress(ctypes.cast(out, ctypes.c_void_p), cap, sp, len(src))
    err = _get_error(written)
    if err is not None:
        raise err
    dst[:written] = out.raw[:written]
    return written


def Decompress(dst, src):
    """zstd.go:134-158.  On "Destination buffer is too small" the reference falls back to its streaming reader
    (zstd.go:154-157), whose result is simply the whole content; here the frame is re-run with a larger buffer."""
    if len(src) == 0:
        raise ErrEmptySlice
    bound = decompressSizeHint(src)
    cap = len(dst) if (dst is not None and len(dst) >= bound) else bound
    while True:
        buf = bytearray(cap)
        try:
            written = DecompressInto(buf, src)
            return bytes(buf[:written])
        except ErrorCode as e:
            if not IsDstSizeTooSmallError(e) or cap > (1 << 33):
                raise
            cap *= 4


def _batch_call(fn, srcs, caps, extra):
    n = len(srcs)
    keep = []
    srcp = (ctypes.c_void_p * n)()
    srcn = (ctypes.c_size_t * n)()
    dstp = (ctypes.c_void_p * n)()
    dstc = (ctypes.c_size_t * n)()
    outn = (ctypes.c_size_t * n)()
    outs = []
    for i, s in enumerate(srcs):
        p, k = _buf(s)
        keep.append(k)
        srcp[i] = p.value if len(s) else None
        srcn[i] = len(s)
        o = (ctypes.c_char * max(caps[i], 1))()
        outs.append(o)
        dstp[i] = ctypes.cast(o, ctypes.c_void_p).value
        dstc[i] = caps[i]
    rc = fn(ctypes.cast(srcp, c_void_pp), ctypes.cast(srcn, c_size_p), ctypes.cast(dstp, c_void_pp),
            ctypes.cast(dstc, c_size_p), ctypes.cast(outn, c_size_p), n, *extra)
    err = _get_error(rc)
    if err is not None:
        raise err
    res = []
    for i in range(n):
        e = _get_error(outn[i])
        res.append(e if e is not None else outs[i].raw[:outn[i]])
    return res


def CompressBatch(srcs, level=BestSpeed, cdict=None):
    """New surface: CompressLevel over many independent payloads in one launch.
    Returns a list whose items are bytes or an ErrorCode (per-frame, like the per-call return of zstd.go:123-128)."""
    return _batch_call(lib.zsa_compress_batch, srcs, [CompressBound(len(s)) for s in srcs], (level, cdict))


def DecompressBatch(srcs, caps=None, ddict=None):
    """New surface: DecompressInto over many frames in one launch; caps default to decompressSizeHint."""
    if caps is None:
        caps = [decompressSizeHint(s) for s in srcs]
    return _batch_call(lib.zsa_decompress_batch, srcs, caps, (ddict,))


class ctx:
    """zstd_ctx.go:38-131 -- the reusable-context flavour of the one-shot API.  libzstd contexts carry nothing from one
    one-shot call to the next that changes the result ("first and later CCtx usages compress the same"), so the engine needs
    no per-context state: the methods run the same entry points (zstd_ctx.go:70,78 ZSTD_compressCCtx -> zsa_compress,
    :107 ZSTD_decompressDCtx -> zsa_decompress) with the same dst-reuse and fallback rules."""

    def Compress(self, dst, src):
        return self.CompressLevel(dst, src, DefaultCompression)        # zstd_ctx.go:53-55

    def CompressLevel(self, dst, src, level):
        return CompressLevel(dst, src, level)                          # zstd_ctx.go:57-91

    def Decompress(self, dst, src):
        return Decompress(dst, src)                                    # zstd_ctx.go:93-126


def NewCtx():
    """zstd_ctx.go:43-51."""
    return ctx()


class BulkProcessor:
    """zstd_bulk.go:27-151."""

    def __init__(self, dictionary, compressionLevel):
        if len(dictionary) < 1:
            raise ErrEmptyDictionary
        p, keep = _buf(dictionary)
        self.cDict = lib.zsa_createCDict(p, len(dictionary), compressionLevel)
        if not self.cDict:
            raise ErrBadDictionary
        self.dDict = lib.zsa_createDDict(p, len(dictionary))
        if not self.dDict:
            lib.zsa_freeCDict(self.cDict)
            self.cDict = None
            raise ErrBadDictionary

    def Compress(self, dst, src):
        """zstd_bulk.go:63-103."""
        bound = CompressBound(len(src))
        out = (ctypes.c_char * max(bound, 1))()
        sp, keep = _buf(src)
        written = lib.zsa_compress_usingCDict(ctypes.cast(out, ctypes.c_void_p), bound, sp if len(src) else None, len(src), self.cDict)
        err = _get_error(written)
        if err is not None:
            raise err
        return out.raw[:written]

    def Decompress(self, dst, src):
        """zstd_bulk.go:108-141."""
        if len(src) == 0:
            raise ErrEmptySlice
        contentSize = decompressSizeHint(src)
        cap = len(dst) if (dst is not None and len(dst) >= contentSize) else contentSize
        if cap == 0:
            return b""
        out = (ctypes.c_char * cap)()
        sp, keep = _buf(src)
        written = lib.zsa_decompress_usingDDict(ctypes.cast(out, ctypes.c_void_p), cap, sp, len(src), self.dDict)
        err = _get_error(written)
        if err is not None:
            raise err
        return out.raw[:written]

    def CompressBatch(self, srcs):
        return CompressBatch(srcs, 0, self.cDict)

    def DecompressBatch(self, srcs, caps=None):
        return DecompressBatch(srcs, caps, self.dDict)

    def __del__(self):   # zstd_bulk.go:144-151 finalizeBulkProcessor
        try:
            if getattr(self, "cDict", None):
                lib.zsa_freeCDict(self.cDict)
            if getattr(self, "dDict", None):
                lib.zsa_freeDDict(self.dDict)
        except Exception:
            pass


def NewBulkProcessor(dictionary, compressionLevel):
    """zstd_bulk.go:33-58."""
    return BulkProcessor(dictionary, compressionLevel)


def getCParams(level, srcSize, dictSize=0, mode=0):
    out = (ctypes.c_uint * 7)()
    if lib.zsa_getCParams(level, srcSize, dictSize, mode, out):
        return None
    return tuple(out)


class DeviceBatch:
    """Device-resident batch (section 3 of include/zstd_amd.h): frames are slices of torch CUDA(HIP) tensors."""

    def __init__(self, maxFrames):
        self.h = lib.zsa_batch_create(maxFrames)
        if not self.h:
            raise ZstdError("zsa_batch_create failed: no usable HIP device or out of device memory")
        self.maxFrames = maxFrames
        self.n = 0

    def set_frames(self, src_ptr, src_offsets, src_sizes, dst_ptr, dst_offsets, dst_caps, stream=None):
        n = len(src_sizes)
        for i in range(n):
            lib.zsa_batch_set_frame(self.h, i, src_ptr + int(src_offsets[i]), int(src_sizes[i]),
                                    dst_ptr + int(dst_offsets[i]), int(dst_caps[i]))
        self.n = n
        rc = lib.zsa_batch_upload(self.h, n, stream)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)

    def set_frames_strided(self, n, src_ptr, src_stride, src_sizes, dst_ptr, dst_stride, dst_cap, stream=None):
        sizes = None
        if src_sizes is not None:
            sizes = (ctypes.c_size_t * n)(*[int(x) for x in src_sizes])
        rc = lib.zsa_batch_set_frames_strided(self.h, n, src_ptr, src_stride, sizes, dst_ptr, dst_stride, dst_cap)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)
        self.n = n
        rc = lib.zsa_batch_upload(self.h, n, stream)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)

    def link_src_sizes(self, producer, stream=None):
        rc = lib.zsa_batch_link_src_sizes(self.h, producer.h, stream)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)

    def compress(self, level=BestSpeed, cdict=None, stream=None):
        rc = lib.zsa_compress_enqueue(self.h, level, cdict, stream)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)

    def decompress(self, ddict=None, stream=None):
        rc = lib.zsa_decompress_enqueue(self.h, ddict, stream)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)

    def results(self, stream=None):
        out = (ctypes.c_size_t * max(self.n, 1))()
        rc = lib.zsa_batch_results(self.h, ctypes.cast(out, c_size_p), stream)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)
        return [out[i] for i in range(self.n)]

    def export_sizes_i32(self, out_ptr, stream=None):
        """result words -> int32 on the device (bytes, or -(error code)), into a caller-owned device buffer"""
        rc = lib.zsa_batch_export_sizes_i32(self.h, out_ptr, stream)
        if lib.zsa_isError(rc):
            raise ErrorCode(rc - 2 ** 64)

    def profile(self, enable, read=False, stream=None):
        out = (ctypes.c_uint64 * 12)() if read else None
        lib.zsa_batch_profile(self.h, 1 if enable else 0, out, stream)
        return list(out) if read else None

    def last_launch(self):
        wg, lds = ctypes.c_uint(), ctypes.c_uint()
        lib.zsa_batch_last_launch(self.h, ctypes.byref(wg), ctypes.byref(lds))
        return wg.value, lds.value

    def close(self):
        if self.h:
            lib.zsa_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
