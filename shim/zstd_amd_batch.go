//go:build zstd_amd && go1.21
// +build zstd_amd,go1.21

package zstd

/*
#include <stdlib.h>
#include "zstd_amd.h"
*/
import "C"
import (
	"errors"
	"runtime"
	"unsafe"
)

// New surface: many independent payloads in one GPU launch (include/zstd_amd.h section 2).  Per payload the result is
// what the single call would have returned -- the same bytes, or the same error -- so a caller that loops over
// CompressLevel / Decompress today (zstd_stream_test.go:149-169 fans such loops out over goroutines) can hand the
// whole slice over instead.

// ErrBatchShape is returned when dst and payloads differ in length.
var ErrBatchShape = errors.New("dst and payloads must have the same length")

// cArrays holds the argument arrays of one batch call in C memory: cgo forbids passing a Go slice of Go pointers,
// so the pointer tables live outside the Go heap and the payloads they point to are pinned for the call.
type cArrays struct {
	n                       int
	src, dst                *unsafe.Pointer
	srcSize, dstCap, result *C.size_t
	pin                     runtime.Pinner
}

func newCArrays(n int) *cArrays {
	a := &cArrays{n: n}
	a.src = (*unsafe.Pointer)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(uintptr(0)))))
	a.dst = (*unsafe.Pointer)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(uintptr(0)))))
	a.srcSize = (*C.size_t)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.size_t(0)))))
	a.dstCap = (*C.size_t)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.size_t(0)))))
	a.result = (*C.size_t)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.size_t(0)))))
	return a
}

func (a *cArrays) set(i int, src, dst []byte) {
	srcs := unsafe.Slice(a.src, a.n)
	dsts := unsafe.Slice(a.dst, a.n)
	if len(src) > 0 {
		a.pin.Pin(&src[0])
		srcs[i] = unsafe.Pointer(&src[0])
	}
	if len(dst) > 0 {
		a.pin.Pin(&dst[0])
		dsts[i] = unsafe.Pointer(&dst[0])
	}
	unsafe.Slice(a.srcSize, a.n)[i] = C.size_t(len(src))
	unsafe.Slice(a.dstCap, a.n)[i] = C.size_t(len(dst))
}

func (a *cArrays) free() {
	a.pin.Unpin()
	C.free(unsafe.Pointer(a.src))
	C.free(unsafe.Pointer(a.dst))
	C.free(unsafe.Pointer(a.srcSize))
	C.free(unsafe.Pointer(a.dstCap))
	C.free(unsafe.Pointer(a.result))
}

func compressBatch(dst, payloads [][]byte, level int, cdict *C.zsa_CDict, engine bool, perCall func(d, s []byte) int) ([][]byte, []error, error) {
	if len(dst) != len(payloads) {
		return nil, nil, ErrBatchShape
	}
	n := len(payloads)
	out := make([][]byte, n)
	errs := make([]error, n)
	if n == 0 {
		return out, errs, nil
	}
	for i := range payloads { // same buffer policy as CompressLevel, zstd.go:96-101
		bound := CompressBound(len(payloads[i]))
		if cap(dst[i]) >= bound {
			dst[i] = dst[i][0:bound]
		} else {
			dst[i] = make([]byte, bound)
		}
	}
	results := make([]int, n)
	if engine {
		a := newCArrays(n)
		defer a.free()
		for i := range payloads {
			a.set(i, payloads[i], dst[i])
		}
		staged := 0
		for i := range payloads {
			staged += len(payloads[i])
		}
		enter()
		rc := int(C.zsa_compress_batch(a.src, a.srcSize, a.dst, a.dstCap, a.result, C.size_t(n), C.int(level), cdict))
		failed := rc < 0 && engineFailed()
		leave(staged)
		res := unsafe.Slice(a.result, n)
		for i := range results {
			results[i] = int(res[i])
			if rc < 0 { // the batch as a whole could not run
				results[i] = rc
			}
			if failed { // ... for a reason of the engine's own (device lost, out of device memory): payload by payload on the CPU
				results[i] = errParameterUnsupported
			}
		}
	} else {
		for i := range results {
			results[i] = errParameterUnsupported
		}
	}
	for i, r := range results {
		if r == errParameterUnsupported { // outside the engine: the vendored path, payload by payload
			r = perCall(dst[i], payloads[i])
		}
		if err := getError(r); err != nil {
			errs[i] = err
			continue
		}
		out[i] = dst[i][:r]
	}
	return out, errs, nil
}

// CompressBatch compresses payloads[i] at `level` into dst[i] (reused when its capacity is at least
// CompressBound(len(payloads[i])), else allocated), all in one GPU launch.  out[i] equals what
// CompressLevel(dst[i], payloads[i], level) returns; errs[i] is that call's error.
func CompressBatch(dst, payloads [][]byte, level int) (out [][]byte, errs []error, err error) {
	return compressBatch(dst, payloads, level, nil, amdUsable, func(d, s []byte) int { return vendoredCompress(d, s, level) })
}

// CompressBatch is CompressBatch with the processor's dictionary: out[i] equals p.Compress(dst[i], payloads[i]).
func (p *BulkProcessor) CompressBatch(dst, payloads [][]byte) (out [][]byte, errs []error, err error) {
	// (a processor whose level the engine does not serve has no device dictionary: every payload takes the vendored path)
	return compressBatch(dst, payloads, 0, p.accel.cDict, p.accel.cDict != nil, func(d, s []byte) int { return hotCompressDict(p, d, s) })
}

func decompressBatch(dst, frames [][]byte, ddict *C.zsa_DDict, engine bool, perCall func(d, s []byte) int) ([][]byte, []error, error) {
	if len(dst) != len(frames) {
		return nil, nil, ErrBatchShape
	}
	n := len(frames)
	out := make([][]byte, n)
	errs := make([]error, n)
	if n == 0 {
		return out, errs, nil
	}
	for i := range frames { // same buffer policy as Decompress, zstd.go:139-144
		if len(frames[i]) == 0 {
			errs[i] = ErrEmptySlice
			continue
		}
		bound := decompressSizeHint(frames[i])
		if cap(dst[i]) >= bound {
			dst[i] = dst[i][0:cap(dst[i])]
		} else {
			dst[i] = make([]byte, bound)
		}
	}
	results := make([]int, n)
	if engine {
		a := newCArrays(n)
		defer a.free()
		for i := range frames {
			if errs[i] == nil {
				a.set(i, frames[i], dst[i])
			}
		}
		staged := 0
		for i := range dst {
			staged += len(dst[i])
		}
		enter()
		rc := int(C.zsa_decompress_batch(a.src, a.srcSize, a.dst, a.dstCap, a.result, C.size_t(n), ddict))
		failed := rc < 0 && engineFailed()
		leave(staged)
		res := unsafe.Slice(a.result, n)
		for i := range results {
			results[i] = int(res[i])
			if rc < 0 {
				results[i] = rc
			}
		}
		if failed { // the engine could not run the batch (device lost, out of device memory): frame by frame on the CPU
			engine = false
		}
	}
	for i, r := range results {
		if errs[i] != nil {
			continue
		}
		if !engine || (r == errPrefixUnknown && legacyMagic(frames[i])) {
			r = perCall(dst[i], frames[i])
		}
		if err := getError(r); err != nil {
			// A frame without a content size whose hint was too small: Decompress (zstd.go:151-159) falls back to the
			// streaming reader.  The batch reports the error; the caller retries that one frame with Decompress.
			errs[i] = err
			continue
		}
		out[i] = dst[i][:r]
	}
	return out, errs, nil
}

// DecompressBatch decompresses frames[i] into dst[i] in one GPU launch.  out[i] equals what DecompressInto would have
// produced in a buffer sized by decompressSizeHint; errs[i] is ErrEmptySlice for an empty frame, or the call's error.
func DecompressBatch(dst, frames [][]byte) (out [][]byte, errs []error, err error) {
	return decompressBatch(dst, frames, nil, amdUsable, vendoredDecompress)
}

// DecompressBatch is DecompressBatch with the processor's dictionary: out[i] equals p.Decompress(dst[i], frames[i]).
func (p *BulkProcessor) DecompressBatch(dst, frames [][]byte) (out [][]byte, errs []error, err error) {
	return decompressBatch(dst, frames, p.accel.dDict, p.accel.dDict != nil, func(d, s []byte) int { return hotDecompressDict(p, d, s) })
}
