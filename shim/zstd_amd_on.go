//go:build zstd_amd
// +build zstd_amd

package zstd

/*
#cgo CFLAGS:  -I${SRCDIR}/third_party/zstd_amd/include
#cgo LDFLAGS: -L${SRCDIR}/third_party/zstd_amd -lzstd_amd -Wl,-rpath,${SRCDIR}/third_party/zstd_amd
#include "zstd.h"
#include "zstd_amd.h"
*/
import "C"
import (
	"encoding/binary"
	"runtime"
	"unsafe"
)

// Hooks of the one-shot / bulk hot path, routed through libzstd_amd.so (include/zstd_amd.h).  Return convention is
// libzstd's: bytes written, or -(ZSTD_ErrorCode); getError (errors.go:21-27) reads both libraries' codes alike,
// because the engine returns the reference's codes and strings.
//
// The engine has no CPU path of its own.  What it does not serve comes back as an error code and is answered by the
// vendored C right here, so callers of the package see one behaviour:
//   -40 "Unsupported parameter"      a level/size whose strategy is btlazy2 or above, or a payload above the
//                                    attach cut-off (16 / 32 KiB) with a dictionary of level 3 and up
//   -10 "Unknown frame descriptor"   on a frame that starts with a legacy (v0.1-v0.7) magic number: the vendored
//                                    build may carry ZSTD_LEGACY_SUPPORT, the engine does not
//   no usable HIP device             amdUsable is false and the hooks never enter the engine
//   any engine-side failure          a HIP call that failed, device or pinned memory that could not be had, a thread that
//                                    changed devices: C.zsa_engine_failure() is non-zero after the call (include/zstd_amd.h
//                                    section 5).  The code that came with it ("Error (generic)", "Allocation error", "Operation
//                                    not authorized") is one the reference's CompressLevel / Decompress cannot produce for these
//                                    arguments (zstd.go:108-128), so the caller never sees it.
const (
	errPrefixUnknown        = -10
	errParameterUnsupported = -40
)

var amdUsable = int(C.zsa_device_count()) > 0

func bytesPtr(b []byte) unsafe.Pointer {
	if len(b) == 0 {
		return unsafe.Pointer(nil) // zstd.go:103-105: never take &b[0] of an empty slice
	}
	return unsafe.Pointer(&b[0])
}

// legacyMagic reports whether src starts with one of the v0.1-v0.7 frame magics (zstd_legacy.h: 0xFD2FB51E, 0xFD2FB522..27).
func legacyMagic(src []byte) bool {
	if len(src) < 4 {
		return false
	}
	m := binary.LittleEndian.Uint32(src)
	return m == 0xFD2FB51E || (m >= 0xFD2FB522 && m <= 0xFD2FB527)
}

// engineFailed reports whether the engine call this goroutine's OS thread just made failed for a reason of the engine's
// own.  The calling goroutine must not have been moved between the call and this query: both run inside one cgo-free
// stretch of a function that holds runtime.LockOSThread (see enter/leave below).
func engineFailed() bool { return C.zsa_engine_failure() != 0 }

// enter pins the goroutine to its OS thread for the span of one engine call plus the query that follows it: the engine's
// failure flag and its staging buffers are per OS thread.
func enter() { runtime.LockOSThread() }

// leave ends the span.  After a large batch the thread's staging buffers (pinned and device memory sized to that batch)
// are handed back: Go's OS threads rarely exit, so nothing else would ever release them.
func leave(stagedBytes int) {
	if stagedBytes >= releaseStagingAbove {
		C.zsa_thread_release()
	}
	runtime.UnlockOSThread()
}

// releaseStagingAbove bounds what an idle OS thread keeps: staging of calls below it (single payloads, small batches) is
// kept for the next call, staging of larger batches is released as the call ends.
const releaseStagingAbove = 64 << 20

func vendoredCompress(dst, src []byte, level int) int {
	return int(C.ZSTD_compress(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), C.int(level)))
}

func vendoredDecompress(dst, src []byte) int {
	return int(C.ZSTD_decompress(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src))))
}

// hotCompress replaces C.ZSTD_compress at zstd.go:108,115 (and C.ZSTD_compressCCtx at zstd_ctx.go:70,78).
func hotCompress(dst, src []byte, level int) int {
	if amdUsable {
		enter()
		r := int(C.zsa_compress(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), C.int(level)))
		failed := r < 0 && engineFailed()
		leave(len(src))
		if r != errParameterUnsupported && !failed {
			return r
		}
	}
	return vendoredCompress(dst, src, level)
}

// hotDecompress replaces C.ZSTD_decompress at zstd.go:168 (and C.ZSTD_decompressDCtx at zstd_ctx.go:107).
func hotDecompress(dst, src []byte) int {
	if amdUsable {
		enter()
		r := int(C.zsa_decompress(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src))))
		failed := r < 0 && engineFailed()
		leave(len(dst))
		if !(r == errPrefixUnknown && legacyMagic(src)) && !failed {
			return r
		}
	}
	return vendoredDecompress(dst, src)
}

// bulkAccel is the engine's half of a BulkProcessor: the dictionary digested into device memory
// (zstd_bulk.go:41-52 digests it for the CPU).  cDict stays nil when the engine does not serve the level
// (zsa_createCDict returns NULL then); the vendored cDict answers in that case.
type bulkAccel struct {
	cDict *C.zsa_CDict
	dDict *C.zsa_DDict
}

func newBulkAccel(dictionary []byte, compressionLevel int) bulkAccel {
	var a bulkAccel
	if !amdUsable || len(dictionary) == 0 {
		return a
	}
	a.cDict = C.zsa_createCDict(unsafe.Pointer(&dictionary[0]), C.size_t(len(dictionary)), C.int(compressionLevel))
	a.dDict = C.zsa_createDDict(unsafe.Pointer(&dictionary[0]), C.size_t(len(dictionary)))
	return a
}

// free is called from finalizeBulkProcessor (zstd_bulk.go:144-151).
func (a *bulkAccel) free() {
	if a.cDict != nil {
		C.zsa_freeCDict(a.cDict)
		a.cDict = nil
	}
	if a.dDict != nil {
		C.zsa_freeDDict(a.dDict)
		a.dDict = nil
	}
}

// hotCompressDict replaces the CCtx create / C.ZSTD_compress_usingCDict / free sequence at zstd_bulk.go:71-96.
func hotCompressDict(p *BulkProcessor, dst, src []byte) int {
	if p.accel.cDict != nil {
		enter()
		r := int(C.zsa_compress_usingCDict(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), p.accel.cDict))
		failed := r < 0 && engineFailed()
		leave(len(src))
		if r != errParameterUnsupported && !failed {
			return r
		}
	}
	cctx := C.ZSTD_createCCtx()
	defer C.ZSTD_freeCCtx(cctx)
	return int(C.ZSTD_compress_usingCDict(cctx, bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), p.cDict))
}

// hotDecompressDict replaces the DCtx create / C.ZSTD_decompress_usingDDict / free sequence at zstd_bulk.go:124-132.
func hotDecompressDict(p *BulkProcessor, dst, src []byte) int {
	if p.accel.dDict != nil {
		enter()
		r := int(C.zsa_decompress_usingDDict(bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), p.accel.dDict))
		failed := r < 0 && engineFailed()
		leave(len(dst))
		if !(r == errPrefixUnknown && legacyMagic(src)) && !failed {
			return r
		}
	}
	dctx := C.ZSTD_createDCtx()
	defer C.ZSTD_freeDCtx(dctx)
	return int(C.ZSTD_decompress_usingDDict(dctx, bytesPtr(dst), C.size_t(len(dst)), bytesPtr(src), C.size_t(len(src)), p.dDict))
}
