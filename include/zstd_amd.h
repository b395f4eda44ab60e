/* zstd_amd.h -- C ABI of libzstd_amd.so: the MI355X-native batch engine behind DataDog/zstd's one-shot and
 * bulk (dictionary) call sites.  Plain pointers and sizes only; no HIP or torch types.
 *
 * Return convention is libzstd's (reference zstd.h / error_private.h:57-61): a size_t that is either a byte
 * count or (size_t)-ZSTD_ErrorCode; zsa_isError()/zsa_getErrorName() give the same answers and strings as the
 * reference's ZSTD_isError()/ZSTD_getErrorName() (errors.go:21-34 string-matches them).
 *
 * Scope of the engine: compression at the levels whose strategy is ZSTD_fast (level 1 and the negative levels at every
 * size, level 2 above 256 KB and up to 128 KB), ZSTD_dfast (levels 3 and 0 at every size, level 4 except at <= 16 KB
 * and in (128 KB, 256 KB], level 2 in (128 KB, 256 KB]) or ZSTD_greedy / ZSTD_lazy / ZSTD_lazy2 (level 5 -- the default of
 * the reference's Compress(), zstd.go:23 -- at every size; levels 4 to 12 wherever clevels.h selects one of the three),
 * byte-identical to libzstd 1.5.5;
 * dictionary compression (CDict) at every one of these strategies -- at the ZSTD_fast levels for payloads of any size, above
 * them (a CDict of level 3 and up) for payloads up to the reference's attach cut-off, 16 KiB at ZSTD_dfast and 32 KiB at
 * ZSTD_greedy .. ZSTD_lazy2 (zstd_compress.c:2241-2252), larger ones answering parameter_unsupported; decompression of any
 * Zstandard frame (concatenated and skippable frames, checksums, dictionaries) except legacy v0.x frames.
 * Anything else returns an error code -- there is no CPU fallback inside this library: every entry point
 * that moves data runs on the GPU and fails loudly when no HIP device is usable.
 */
#ifndef ZSTD_AMD_H
#define ZSTD_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------------------
 * 1. Single-call entry points -- what the reference's cgo stubs bind today.  Host pointers, synchronous.
 *    Each is a batch of one frame through the same kernels as section 2.
 * ------------------------------------------------------------------------------------------------------- */

/* replaces C.ZSTD_compressBound (zstd.go:57); pure arithmetic, zstd.h ZSTD_COMPRESSBOUND */
size_t zsa_compressBound(size_t srcSize);
/* replaces C.ZSTD_compress (zstd.go:108,115) and C.ZSTD_compressCCtx (zstd_ctx.go:70,78) */
size_t zsa_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
/* replaces C.ZSTD_decompress (zstd.go:168) and C.ZSTD_decompressDCtx (zstd_ctx.go:107) */
size_t zsa_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
/* replaces C.ZSTD_getFrameContentSize (zstd.go:71); header peek on the host, zstd_decompress.c:565-581 */
unsigned long long zsa_getFrameContentSize(const void* src, size_t srcSize);
/* replace C.ZSTD_isError / C.ZSTD_getErrorName (errors.go:13,22) */
unsigned    zsa_isError(size_t code);
const char* zsa_getErrorName(size_t code);

/* dictionaries: replace C.ZSTD_createCDict / ZSTD_freeCDict / ZSTD_createDDict / ZSTD_freeDDict
 * (zstd_bulk.go:41-52,144-151), C.ZSTD_compress_usingCDict (zstd_bulk.go:77,86) and
 * C.ZSTD_decompress_usingDDict (zstd_bulk.go:125).  The digested tables live in device memory. */
typedef struct zsa_CDict_s zsa_CDict;
typedef struct zsa_DDict_s zsa_DDict;
zsa_CDict* zsa_createCDict(const void* dict, size_t dictSize, int compressionLevel);
size_t     zsa_freeCDict(zsa_CDict* cdict);
zsa_DDict* zsa_createDDict(const void* dict, size_t dictSize);
size_t     zsa_freeDDict(zsa_DDict* ddict);
size_t zsa_compress_usingCDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zsa_CDict* cdict);
size_t zsa_decompress_usingDDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zsa_DDict* ddict);

/* ---------------------------------------------------------------------------------------------------------
 * 2. Batch entry points, host memory -- many independent frames in one launch.  New surface (the single-call
 *    ABI cannot express it); semantics per frame are exactly those of the calls in section 1.
 *    outSize[i] receives bytes written or (size_t)-error for frame i.  Returns 0, or an error code when the
 *    batch as a whole could not run (no device, allocation failure).
 *    Go side: CompressBatch / (*BulkProcessor).CompressBatch in INTEGRATION.md.
 *    Batches of 8 MiB and more are staged by helper threads that live for the call (ZSTD_AMD_COPY_THREADS,
 *    default 8, 1 = none; they copy between the caller's buffers and pinned memory and make no HIP calls) and run as up
 *    to four sub-batches on streams of their own, so that upload, kernel and download overlap.
 *    ZSTD_AMD_DEVICES=<n|all> (default 1: the calling thread's current device only) spreads a batch of 64 MiB and more that
 *    uses no dictionary over n GPUs -- contiguous shares of about equal bytes, one device-bound worker thread each; the
 *    reference's callers reach the same by fanning out over goroutines (zstd_stream_test.go:149-169).
 * ------------------------------------------------------------------------------------------------------- */
size_t zsa_compress_batch(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity,
                          size_t* outSize, size_t nFrames, int compressionLevel, const zsa_CDict* cdict /* or NULL */);
size_t zsa_decompress_batch(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity,
                            size_t* outSize, size_t nFrames, const zsa_DDict* ddict /* or NULL */);

/* ---------------------------------------------------------------------------------------------------------
 * 3. Batch entry points, device memory -- inputs and outputs already resident in HBM (what bench.py times).
 *    A zsa_batch owns everything a launch needs (frame table, result words, per-workgroup scratch), so
 *    zsa_*_enqueue only put work on `stream` (a hipStream_t passed as void*; NULL = the default stream) and allocate
 *    nothing -- except a zsa_compress_enqueue that needs a larger compression workspace than the batch holds so far
 *    (normally only the first: the workspace is sized from the launch -- largest block and largest match tables of its
 *    frames, e.g. 0.6 MiB per resident workgroup for 128 KiB level-1 frames).  A batch, the dictionaries used with it and
 *    all device pointers must belong to the HIP device that was current at zsa_batch_create; an enqueue made under
 *    another device returns "Operation not authorized at current processing stage".
 *    Capacity contract of the DEVICE-memory entry points: dstCapacity >= zsa_compressBound(srcSize) -- what every reference
 *    call site passes (zstd.go:96-101) -- else "Destination buffer is too small", even where the frame would have fit.
 *    The HOST-memory entry points (sections 1 and 2) take any capacity, as ZSTD_compress does: the frame is produced in a
 *    bound-sized device slot and handed over when it fits the caller's buffer, else "Destination buffer is too small".
 *    Frames whose sizes are linked on the device (zsa_batch_link_src_sizes) are sized for match tables of up to 8 MiB;
 *    a frame that needs more (levels 9 and up on inputs above 256 KB) gets "Unsupported parameter" in its result word.
 *    A compress launch whose workspace does not fit half of the device's free memory keeps fewer frames in flight, and an
 *    allocation that is still refused is tried once more at half that before "Allocation error" is returned.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct zsa_batch_s zsa_batch;
zsa_batch* zsa_batch_create(size_t maxFrames);
void       zsa_batch_free(zsa_batch* b);
/* describe frame i: device pointers */
size_t zsa_batch_set_frame(zsa_batch* b, size_t i, const void* d_src, size_t srcSize, void* d_dst, size_t dstCapacity);
/* describe frames 0..nFrames-1 laid out at fixed strides: frame i reads d_src + i*srcStride (srcSizes[i] bytes, or
 * srcStride bytes when srcSizes is NULL) and writes d_dst + i*dstStride (capacity dstCapacity each) */
size_t zsa_batch_set_frames_strided(zsa_batch* b, size_t nFrames, const void* d_src, size_t srcStride, const size_t* srcSizes,
                                    void* d_dst, size_t dstStride, size_t dstCapacity);
/* on the device, without a host round trip: consumer frame i takes its srcSize from the result word of producer
 * frame i (a failed producer frame yields srcSize 0).  Lets compress -> decompress chains stay on the GPU. */
size_t zsa_batch_link_src_sizes(zsa_batch* consumer, const zsa_batch* producer, void* stream);
/* upload the first nFrames descriptors (async on stream) */
size_t zsa_batch_upload(zsa_batch* b, size_t nFrames, void* stream);
/* launch; results stay on the device until zsa_batch_results */
size_t zsa_compress_enqueue(zsa_batch* b, int compressionLevel, const zsa_CDict* cdict, void* stream);
size_t zsa_decompress_enqueue(zsa_batch* b, const zsa_DDict* ddict, void* stream);
/* wait for `stream` and copy the per-frame results (bytes or -(error)) to outSize[0..nFrames) */
size_t zsa_batch_results(zsa_batch* b, size_t* outSize, void* stream);
/* device pointer to the per-frame result words (uint64), for consumers that stay on the GPU */
const uint64_t* zsa_batch_device_results(const zsa_batch* b);
/* on the device: result words -> int32 (bytes written, or -(error code)) into d_out[0..nFrames): the tensor a rank hands to
 * the all-gather of compressed sizes that gives every rank the global output offsets (no reference counterpart:
 * the reference's callers fan out over goroutines, zstd_stream_test.go:149-169) */
size_t zsa_batch_export_sizes_i32(const zsa_batch* b, int32_t* d_out, void* stream);
/* geometry of the last launch (for roofline bookkeeping): workgroups launched, LDS bytes per workgroup */
void zsa_batch_last_launch(const zsa_batch* b, unsigned* workgroups, unsigned* ldsBytes);

/* diagnostics: enable/disable per-phase cycle accounting for launches of this batch; when `out` is non-NULL it first
 * receives the 12 accumulated counters (shader-clock cycles of lane 0, summed over workgroups). */
size_t zsa_batch_profile(zsa_batch* b, int enable, uint64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * 4. One entry point that allocates nothing.  No counterpart in the reference: it is for callers that cannot call
 *    hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 *    Decompresses nFrames frames described by a DEVICE array of zsa_frame records (device pointers inside), writing one
 *    result word per frame (bytes or (uint64)-error, as zsa_batch_device_results) to d_results.  Same kernel as
 *    zsa_decompress_enqueue; its workspace (work counter, per-workgroup literal buffer and parked tables) is a
 *    19 MB array inside the library's code object instead of batch-owned memory, each GPU holding its own copy.
 *    Limits: at most 2^20 frames per call (more: "Parameter is out of bound"); 128 resident workgroups, so throughput
 *    is below a zsa_batch's.  Calls on one GPU share that array: calls that overlap on one GPU race -- serialise them
 *    (one stream, or events).  The launch is asynchronous on `stream`.
 *    The array's address on a device is looked up once (a runtime query, not a stream operation): by zsa_batch_create,
 *    zsa_createDDict or zsa_noalloc_prepare on that device, else by the first call itself -- so a caller whose first call
 *    would fall under graph capture makes one of those three first, outside the capture.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct { const void* src; void* dst; uint64_t srcSize; uint64_t dstCapacity; } zsa_frame;
size_t zsa_decompress_frames_noalloc(const zsa_frame* d_frames, uint64_t* d_results, size_t nFrames,
                                     const zsa_DDict* ddict /* or NULL */, void* stream);
/* resolves the workspace address on the current device (see above); 0 or an error code.  No counterpart in the reference. */
size_t zsa_noalloc_prepare(void);

/* ---------------------------------------------------------------------------------------------------------
 * 5. What a binding needs around the data path.  No counterparts in libzstd (which has no device to lose).
 * ------------------------------------------------------------------------------------------------------- */
/* Why the LAST data-path call made on this thread failed, when the cause lies with the engine and not with the caller's
 * data: 0 = it did not (the returned code, if any, is the answer libzstd would give), 1 = no usable HIP device, 2 = a HIP
 * runtime call failed, 3 = device / pinned memory could not be allocated, 4 = batch or dictionary used under another device.
 * The codes returned in those cases ("Error (generic)", "Allocation error : not enough memory", "Operation not authorized
 * at current processing stage") are ones libzstd's one-shot calls cannot produce for the same arguments, so a binding answers
 * them with its CPU path: shim/zstd_amd_on.go falls through to the vendored C whenever this is non-zero.  Covers
 * zsa_compress*, zsa_decompress*, zsa_*_batch, zsa_*_enqueue, zsa_create?Dict, zsa_batch_create, zsa_host_alloc. */
int zsa_engine_failure(void);
/* Page-locked host memory.  A large host batch (section 2) whose payloads all lie in such memory is uploaded straight from
 * it, and a decompress batch whose destinations all lie in it is downloaded straight into it -- one DMA copy per run of
 * frames that lie one behind the other -- with no staging copy and no helper threads on that side.  The rates are those of
 * the staged path (PCIe bounds both); what is saved is the host's copy work.  (A Go binding keeps payload slabs in it: C
 * memory, outside the Go heap.)  zsa_host_alloc returns NULL on failure. */
void* zsa_host_alloc(size_t bytes);
void  zsa_host_free(void* p);

/* staging buffers of the single-call / host-batch entry points are per calling thread and released when the thread ends;
 * a long-lived thread can hand them back early.  (libzstd: ZSTD_freeCCtx / ZSTD_freeDCtx, zstd_ctx.go:43-51) */
void zsa_thread_release(void);

/* resolved compression parameters {windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy}
 * for (level, srcSize, dictSize); mirrors ZSTD_getCParams_internal (zstd_compress.c:6960-6983) for the rows this
 * engine implements.  Returns 0, or 1 if the level is outside them.  mode: 0 noAttachDict, 1 attachDict, 2 createCDict. */
int zsa_getCParams(int compressionLevel, unsigned long long srcSize, size_t dictSize, int mode, unsigned out[7]);

/* library / device info */
int         zsa_device_count(void);
const char* zsa_version(void);

#ifdef __cplusplus
}
#endif
#endif
