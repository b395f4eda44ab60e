"""zstd_amd -- MI355X-native batch Zstandard engine behind DataDog/zstd's one-shot and bulk API.

Host-side mirror of the reference's Go surface (zstd.go, zstd_bulk.go, errors.go) over the C ABI of
libzstd_amd.so.  Names, argument meaning and error behaviour follow the Go package.
"""
from .api import (  # noqa: F401
    BestSpeed, BestCompression, DefaultCompression,
    ErrEmptySlice, ErrEmptyDictionary, ErrBadDictionary, ErrorCode,
    CompressBound, Compress, CompressLevel, Decompress, DecompressInto,
    IsDstSizeTooSmallError, BulkProcessor, NewBulkProcessor, NewCtx,
    CompressBatch, DecompressBatch, DeviceBatch, getCParams,
)
