//go:build !zstd_amd
// +build !zstd_amd

package zstd

/*
#include "zstd.h"
*/
import "C"
import "unsafe"

// Hooks of the one-shot / bulk hot path.  Without the zstd_amd tag they are the vendored calls, unchanged:
// zstd.go:108,115 (hotCompress), zstd.go:168 (hotDecompress), zstd_bulk.go:71-96 (hotCompressDict),
// zstd_bulk.go:124-132 (hotDecompressDict).  All return libzstd's size_t convention as an int.

type bulkAccel struct{}

func newBulkAccel(dictionary []byte, compressionLevel int) bulkAccel { return bulkAccel{} }
func (a *bulkAccel) free()                                          {}

func bytesPtr(b []byte) unsafe.Pointer {
	if len(b) == 0 {
		return unsafe.Pointer(nil) // zstd.go:103-105: never take &b[0] of an empty slice
	}
	return unsafe.Pointer(&b[0])
}

func hotCompress(dst, src []byte, level int) int {
	return int(C.ZSTD_compress(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), C.int(level)))
}

func hotDecompress(dst, src []byte) int {
	return int(C.ZSTD_decompress(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src))))
}

func hotCompressDict(p *BulkProcessor, dst, src []byte) int {
	cctx := C.ZSTD_createCCtx()
	defer C.ZSTD_freeCCtx(cctx)
	return int(C.ZSTD_compress_usingCDict(cctx, bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), p.cDict))
}

func hotDecompressDict(p *BulkProcessor, dst, src []byte) int {
	dctx := C.ZSTD_createDCtx()
	defer C.ZSTD_freeDCtx(dctx)
	return int(C.ZSTD_decompress_usingDDict(dctx, bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), p.dDict))
}
