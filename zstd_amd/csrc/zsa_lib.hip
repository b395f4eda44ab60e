// zsa_lib.hip -- libzstd_amd.so: kernels + the C ABI declared in include/zstd_amd.h.
// MI355X (gfx950) only.  No CPU data path: every entry point that moves frame data launches the kernels below.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "zsa_common.h"
#include "zsa_params.h"
#include "zsa_decode.h"
#include "zsa_encode.h"
#include "zsa_dict.h"
#include "../../include/zstd_amd.h"

// ------------------------------------------------------------------------------------------------ kernels
// One 64-lane workgroup per frame at a time; workgroups pull frame indices from a device counter, so a launch
// of (CUs x resident workgroups) keeps every CU busy however frame costs vary.
#ifndef ZSA_DEC_WAVES
#define ZSA_DEC_WAVES 3       // waves per SIMD the decompress kernel is register-budgeted for (12 workgroups per CU)
#endif
#ifndef ZSA_ENC_WAVES
#define ZSA_ENC_WAVES 4       // waves per SIMD the compress kernel is register-budgeted for (16 workgroups per CU)
#endif
#ifndef ZSA_HT_BYTES
#define ZSA_HT_BYTES   ((size_t)64 << 20)                    // match table(s) of the frame in flight: 4 << hashLog bytes, plus
                                                            // 4 << chainLog for ZSTD_dfast (level 4 at > 256 KB: 1 MiB + 1 MiB) or the
                                                            // hash chain; the row finder's records for greedy/lazy/lazy2 take
                                                            // 8 << hashLog (levels 7-8 at > 256 KB: 8 MiB; level 12 at >= 4 MiB:
                                                            // 64 MiB).  A launch is sized to what its frames need, not to this ceiling
#endif
#define ZSA_HT_BYTES_UNSIZED ((size_t)8 << 20)               // ... when the frames' sizes are known on the device only
#define ZSA_PARK_BYTES ((3 * sizeof(ZsaFseCT) + 255) / 256 * 256)   // sequence tables of the last compressed block (ZSTD_lazy and up)
// Compression workspace of one resident workgroup, sized from the launch: `blockCap` = largest block of the batch (<= 128 KiB),
// `htBytes` = largest match-table footprint any of its frames needs.  [seq | lit | tmp | park | tables]
struct ZsaEncLayout { u32 blockCap; u32 litOff, tmpOff, parkOff, htOff; u64 htBytes; u64 stride; };
static ZsaEncLayout zsa_enc_layout(u64 maxSrcSize, u64 htBytes)
{
    ZsaEncLayout L;
    u64 bc = maxSrcSize < ZSA_BLOCKSIZE_MAX ? (maxSrcSize + 255) / 256 * 256 : ZSA_BLOCKSIZE_MAX;
    if (bc < 256) bc = 256;
    L.blockCap = (u32)bc;
    L.litOff = (u32)((bc / 4 + 64) * 8);
    L.tmpOff = L.litOff + (u32)bc + 128;
    L.parkOff = (L.tmpOff + (u32)(bc * 3 / 2) + 512 + 255) / 256 * 256;
    L.htOff = L.parkOff + (u32)ZSA_PARK_BYTES;
    L.htBytes = (htBytes + 255) / 256 * 256;
    L.stride = (u64)L.htOff + L.htBytes;
    return L;
}
#define ZSA_DEC_SCRATCH_BYTES ((size_t)((ZSA_DEC_WKSP_BYTES + 255) / 256 * 256))

struct ZsaLaunchDict { const ZsaCDictDev* cd; const ZsaDDictDev* dd; };

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZSA_DEC_WAVES, ZSA_DEC_WAVES)))
zsa_k_decompress(const ZsaFrameDesc* __restrict__ frames, u64* __restrict__ results, u32 nFrames,
                 u32* __restrict__ counter, u8* __restrict__ scratch, const ZsaDDictDev* dd, u64* __restrict__ prof)
{
    __shared__ ZsaDecShared S;
    __shared__ u32 nextFrame;
    if (threadIdx.x == 0) { for (int i = 0; i < ZSA_NPROF; i++) S.prof[i] = 0; S.profLast = wv_clock(); }
    u8* const litBuf = scratch + (size_t)blockIdx.x * ZSA_DEC_SCRATCH_BYTES;
    for (;;) {
        if (threadIdx.x == 0) nextFrame = atomicAdd(counter, 1u);
        __syncthreads();
        u32 const f = nextFrame;
        __syncthreads();
        if (f >= nFrames) break;
        ZsaFrameDesc const d = frames[f];
        ZSA_PROF(&S, threadIdx.x, 11);
        u64 const r = zsa_decode_frames(&S, d.src, d.srcSize, d.dst, d.dstCapacity, litBuf, dd);
        if (threadIdx.x == 0) results[f] = r;
        __syncthreads();
    }
    if (prof && threadIdx.x == 0) for (int i = 0; i < ZSA_NPROF; i++) atomicAdd((unsigned long long*)&prof[i], (unsigned long long)S.prof[i]);
}

// One persistent compress kernel per parser family (zsa_encode.h).  A launch runs the kernels of the families its level can
// resolve to (clevels.h chooses the strategy per frame by its size: level 2 mixes ZSTD_fast and ZSTD_dfast rows, level 4
// ZSTD_dfast and ZSTD_greedy): each kernel then walks all frames with a counter of its own and takes the ones of its family.
// `famMask` = families launched; the kernel with `reporter` set writes the error word of a frame none of them can take.
template <int FAMILY>
ZSA_DEV void zsa_compress_body(ZsaEncShared& S, u32& nextFrame, const ZsaFrameDesc* __restrict__ frames, u64* __restrict__ results, u32 nFrames,
                               u32* __restrict__ counter, u8* __restrict__ scratch, const ZsaEncLayout L, int level, const ZsaCDictDev* cd,
                               int famMask, int reporter, u64* __restrict__ prof)
{
    if (threadIdx.x == 0) { for (int i = 0; i < ZSA_NPROF; i++) S.prof[i] = 0; S.profLast = wv_clock(); }
    u8* const wg = scratch + (size_t)blockIdx.x * L.stride;
    ZsaEncScratch W;
    W.seq = (u64*)wg; W.lit = wg + L.litOff; W.tmp = wg + L.tmpOff;
    W.park = (ZsaFseCT*)(wg + L.parkOff);
    u32* const zsa_hashTable = (u32*)(wg + L.htOff);
    for (;;) {
        if (threadIdx.x == 0) nextFrame = atomicAdd(counter, 1u);
        __syncthreads();
        u32 const f = nextFrame;
        __syncthreads();
        if (f >= nFrames) break;
        ZsaFrameDesc const d = frames[f];
        ZSA_PROF(&S, threadIdx.x, 11);
        ZsaCParams cp;
        int dictMode = zsa_dm_none;
        int const bad = cd ? zsa_resolveCDictUse(&cp, &dictMode, cd->cp, cd->level, cd->dictSize, d.srcSize)
                           : zsa_resolveCParams(&cp, level, d.srcSize, 0, zsa_cpm_noAttachDict);
        // strategies: ZSTD_fast, and ZSTD_dfast .. ZSTD_lazy2 without a dictionary; the tables must fit the workgroup's scratch
        // (with a dictionary the finder of greedy .. lazy2 is the CDict's, zstd_compress.c:2294: sized for the larger of the two)
        size_t const tableBytes = bad ? 0 : ((size_t)4 << cp.hlog) + (cd && cp.strat >= 2 ? ((size_t)4 << (cp.hlog > cp.clog ? cp.hlog : cp.clog))
                                                                                        : zsa_auxTableBytes(cp.strat, cp.wlog, cp.hlog, cp.clog));
        // (a frame larger than the launch was sized for -- sizes linked on the device -- is refused rather than overrun)
        // with a dictionary above ZSTD_fast: the attached dictionary only (payloads up to 16 KiB at ZSTD_dfast, 32 KiB at greedy .. lazy2)
        bool const unsupported = bad || !(cp.strat == 1 || (cp.strat >= 2 && cp.strat <= 5 && (!cd || (dictMode == zsa_dm_attach && cd->cp.strat == cp.strat)))) || tableBytes > L.htBytes
                              || (d.srcSize > L.blockCap && L.blockCap < ZSA_BLOCKSIZE_MAX);
        int const fam = cd ? (cp.strat == 1 ? 0 : 4) : (cp.strat == 1 ? 2 : (cp.strat == 2 ? 3 : 1));
        if (unsupported || !((famMask >> fam) & 1)) {
            if (reporter && threadIdx.x == 0) results[f] = ZSA_ERR(ZSA_parameter_unsupported);
        } else if (fam == FAMILY) {
            ZsaEncParams P; P.wlog = cp.wlog; P.hlog = cp.hlog; P.mml = cp.mml; P.tlen = cp.tlen; P.clog = cp.clog; P.strat = cp.strat; P.slog = cp.slog;
            bool const withDict = FAMILY == 0 || FAMILY == 4;
            u64 const r = zsa_encode_frame<FAMILY>(&S, zsa_hashTable, P, d.src, d.srcSize, d.dst, d.dstCapacity, W, withDict ? cd : (const ZsaCDictDev*)nullptr, withDict ? dictMode : (int)zsa_dm_none);
            if (threadIdx.x == 0) results[f] = r;
        }
        __syncthreads();
    }
    if (prof && threadIdx.x == 0) for (int i = 0; i < ZSA_NPROF; i++) atomicAdd((unsigned long long*)&prof[i], (unsigned long long)S.prof[i]);
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZSA_ENC_WAVES, ZSA_ENC_WAVES)))
zsa_k_compress(const ZsaFrameDesc* __restrict__ frames, u64* __restrict__ results, u32 nFrames,
               u32* __restrict__ counter, u8* __restrict__ scratch, const ZsaEncLayout L, int level, int famMask, int reporter,
               u64* __restrict__ prof)
{   // ZSTD_fast without a dictionary: the wave parser with the chunk path (zsa_parse.h)
    __shared__ ZsaEncShared S;
    __shared__ u32 nextFrame;
    zsa_compress_body<2>(S, nextFrame, frames, results, nFrames, counter, scratch, L, level, nullptr, famMask, reporter, prof);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZSA_ENC_WAVES, ZSA_ENC_WAVES)))
zsa_k_compress_dict(const ZsaFrameDesc* __restrict__ frames, u64* __restrict__ results, u32 nFrames,
                    u32* __restrict__ counter, u8* __restrict__ scratch, const ZsaEncLayout L, int level, const ZsaCDictDev* cd, int famMask, int reporter,
                    u64* __restrict__ prof)
{   // ZSTD_fast with a dictionary (BulkProcessor)
    __shared__ ZsaEncShared S;
    __shared__ u32 nextFrame;
    zsa_compress_body<0>(S, nextFrame, frames, results, nFrames, counter, scratch, L, level, cd, famMask, reporter, prof);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZSA_ENC_WAVES, ZSA_ENC_WAVES)))
zsa_k_compress_dict_hi(const ZsaFrameDesc* __restrict__ frames, u64* __restrict__ results, u32 nFrames,
                       u32* __restrict__ counter, u8* __restrict__ scratch, const ZsaEncLayout L, int level, const ZsaCDictDev* cd, int famMask, int reporter,
                       u64* __restrict__ prof)
{   // ZSTD_dfast .. ZSTD_lazy2 with an attached dictionary (BulkProcessor at levels 3 and up)
    __shared__ ZsaEncShared S;
    __shared__ u32 nextFrame;
    zsa_compress_body<4>(S, nextFrame, frames, results, nFrames, counter, scratch, L, level, cd, famMask, reporter, prof);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZSA_ENC_WAVES, ZSA_ENC_WAVES)))
zsa_k_compress_dfast(const ZsaFrameDesc* __restrict__ frames, u64* __restrict__ results, u32 nFrames,
                     u32* __restrict__ counter, u8* __restrict__ scratch, const ZsaEncLayout L, int level, int famMask, int reporter,
                     u64* __restrict__ prof)
{   // ZSTD_dfast
    __shared__ ZsaEncShared S;
    __shared__ u32 nextFrame;
    zsa_compress_body<3>(S, nextFrame, frames, results, nFrames, counter, scratch, L, level, nullptr, famMask, reporter, prof);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZSA_ENC_WAVES, ZSA_ENC_WAVES)))
zsa_k_compress_lazy(const ZsaFrameDesc* __restrict__ frames, u64* __restrict__ results, u32 nFrames,
                    u32* __restrict__ counter, u8* __restrict__ scratch, const ZsaEncLayout L, int level, int famMask, int reporter,
                    u64* __restrict__ prof)
{   // ZSTD_greedy / ZSTD_lazy / ZSTD_lazy2
    __shared__ ZsaEncShared S;
    __shared__ u32 nextFrame;
    zsa_compress_body<1>(S, nextFrame, frames, results, nFrames, counter, scratch, L, level, nullptr, famMask, reporter, prof);
}

__global__ void zsa_k_link_sizes(ZsaFrameDesc* __restrict__ frames, const u64* __restrict__ results, u32 n)
{
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u64 const r = results[i]; frames[i].srcSize = (r > ZSA_ERR(ZSA_maxCode)) ? 0 : r; }
}

// result words -> int32 (bytes, or -(error code)): the tensor a rank contributes to the all-gather of compressed sizes
__global__ void zsa_k_export_sizes(const u64* __restrict__ results, s32* __restrict__ out, u32 n)
{
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u64 const r = results[i]; out[i] = (r > ZSA_ERR(ZSA_maxCode)) ? -(s32)(u32)((u64)0 - r) : (s32)r; }
}

// ------------------------------------------------------------------------------------------------ host side
// Engine-side failures -- no usable device, a HIP call that failed, device memory that could not be had, a batch used under
// another device -- are not answers about the caller's data: libzstd on the CPU cannot produce them.  They are returned as
// the nearest libzstd code and, in addition, flagged per calling thread, so that a binding can tell them from data errors
// and answer with its CPU path (zsa_engine_failure, include/zstd_amd.h section 5; shim/zstd_amd_on.go).
enum { ZSA_EF_NONE = 0, ZSA_EF_NO_DEVICE = 1, ZSA_EF_RUNTIME = 2, ZSA_EF_MEMORY = 3, ZSA_EF_DEVICE_MISMATCH = 4 };
static thread_local int t_engineFailure = ZSA_EF_NONE;
static size_t zsa_fail(int why, size_t code) { t_engineFailure = why; return code; }
extern "C" int zsa_engine_failure(void) { return t_engineFailure; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libzstd_amd: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); (void)hipGetLastError(); return zsa_fail(ZSA_EF_RUNTIME, ZSA_ERR(ZSA_GENERIC)); } } while (0)

static const size_t kErrMemory = (size_t)ZSA_ERR(ZSA_memory_allocation);
static const size_t kErrGeneric = (size_t)ZSA_ERR(ZSA_GENERIC);
// test knob: ZSTD_AMD_FAIL=alloc makes every workspace / staging allocation of the data path fail (tests/abi_smoke.c)
static bool zsa_forced_alloc_failure()
{
    static int const on = [] { const char* e = getenv("ZSTD_AMD_FAIL"); return (e && strcmp(e, "alloc") == 0) ? 1 : 0; }();
    return on != 0;
}

struct zsa_batch_s {
    size_t maxFrames, nFrames;
    int  device;                 // HIP device the batch and everything it points to live on (checked at every enqueue)
    ZsaFrameDesc* h_frames;      // pinned
    ZsaFrameDesc* d_frames;
    u64* d_results; u64* h_results;   // h_results pinned
    u32* d_counter;
    u8*  d_scratch; size_t scratchBytes;          // decode workspace (literal buffer + parked tables), per resident workgroup
    u8*  d_encScratch; size_t encScratchBytes;    // compression workspace (ZsaEncLayout per resident workgroup): sized by the launches
    size_t wgs;
    int  numCU;
    u64  maxSrcSize;
    bool sizesOnDevice;          // srcSize words were rewritten on the device (zsa_batch_link_src_sizes): the host copy is stale
    // what a compress launch at (planLevel, planDict) needs (zsa_plan_compress)
    int  planLevel; const void* planDict; bool planValid; u64 planHtBytes;
    unsigned lastWG, lastLDS;
    u64* d_prof; int profOn;
};

extern "C" int zsa_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" const char* zsa_version(void) { return "zstd_amd 0.1 (format-compatible with libzstd 1.5.5, gfx950)"; }

extern "C" size_t zsa_compressBound(size_t srcSize)
{
    if (srcSize >= 0xFF00FF00FF00FF00ULL) return (size_t)ZSA_ERR(ZSA_srcSize_wrong);
    return (size_t)zsa_bound(srcSize);
}
extern "C" unsigned zsa_isError(size_t code) { return code > (size_t)ZSA_ERR(ZSA_maxCode); }
extern "C" const char* zsa_getErrorName(size_t code)
{   // error_private.c:16-66: the strings are part of the boundary (errors.go:29-34 string-matches one of them)
    if (!zsa_isError(code)) return "No error detected";
    switch ((u32)((size_t)0 - code)) {
    case ZSA_GENERIC: return "Error (generic)";
    case ZSA_prefix_unknown: return "Unknown frame descriptor";
    case ZSA_version_unsupported: return "Version not supported";
    case ZSA_frameParameter_unsupported: return "Unsupported frame parameter";
    case ZSA_frameParameter_windowTooLarge: return "Frame requires too much memory for decoding";
    case ZSA_corruption_detected: return "Data corruption detected";
    case ZSA_checksum_wrong: return "Restored data doesn't match checksum";
    case ZSA_literals_headerWrong: return "Header of Literals' block doesn't respect format specification";
    case ZSA_parameter_unsupported: return "Unsupported parameter";
    case ZSA_parameter_combination_unsupported: return "Unsupported combination of parameters";
    case ZSA_parameter_outOfBound: return "Parameter is out of bound";
    case ZSA_init_missing: return "Context should be init first";
    case ZSA_memory_allocation: return "Allocation error : not enough memory";
    case ZSA_workSpace_tooSmall: return "workSpace buffer is not large enough";
    case ZSA_stage_wrong: return "Operation not authorized at current processing stage";
    case ZSA_tableLog_tooLarge: return "tableLog requires too much memory : unsupported";
    case ZSA_maxSymbolValue_tooLarge: return "Unsupported max Symbol Value : too large";
    case ZSA_maxSymbolValue_tooSmall: return "Specified maxSymbolValue is too small";
    case ZSA_stabilityCondition_notRespected: return "pledged buffer stability condition is not respected";
    case ZSA_dictionary_corrupted: return "Dictionary is corrupted";
    case ZSA_dictionary_wrong: return "Dictionary mismatch";
    case ZSA_dictionaryCreation_failed: return "Cannot create Dictionary from provided samples";
    case ZSA_dstSize_tooSmall: return "Destination buffer is too small";
    case ZSA_srcSize_wrong: return "Src size is incorrect";
    case ZSA_dstBuffer_null: return "Operation on NULL destination buffer";
    case ZSA_noForwardProgress_destFull: return "Operation made no progress over multiple calls, due to output buffer being full";
    case ZSA_noForwardProgress_inputEmpty: return "Operation made no progress over multiple calls, due to input being empty";
    case ZSA_frameIndex_tooLarge: return "Frame index is too large";
    case ZSA_seekableIO: return "An I/O error occurred when reading/seeking";
    case ZSA_dstBuffer_wrong: return "Destination buffer is wrong";
    case ZSA_srcBuffer_wrong: return "Source buffer is wrong";
    case ZSA_sequenceProducer_failed: return "Block-level external sequence producer returned an error code";
    case ZSA_externalSequences_invalid: return "External sequences are not valid";
    default: return "Unspecified error code";
    }
}

extern "C" int zsa_getCParams(int level, unsigned long long srcSize, size_t dictSize, int mode, unsigned out[7])
{
    ZsaCParams cp;
    if (zsa_resolveCParams(&cp, level, srcSize, dictSize, mode)) return 1;
    out[0] = cp.wlog; out[1] = cp.clog; out[2] = cp.hlog; out[3] = cp.slog; out[4] = cp.mml; out[5] = cp.tlen; out[6] = cp.strat;
    return 0;
}

// zstd_decompress.c:445-581: frame-header peek done where the bytes are (host memory); no frame data is decoded here
extern "C" unsigned long long zsa_getFrameContentSize(const void* srcv, size_t srcSize)
{
    const u8* src = (const u8*)srcv;
    unsigned long long const ERRV = (unsigned long long)0 - 2, UNKNOWN = (unsigned long long)0 - 1;
    if (srcSize < 5 || src == NULL) return ERRV;
    u32 magic; memcpy(&magic, src, 4);
    if (magic != ZSA_MAGIC) {
        if ((magic & ZSA_MAGIC_SKIP_MASK) == ZSA_MAGIC_SKIP_START) return srcSize < 8 ? ERRV : 0;
        return ERRV;   // includes legacy (v0.x) magics: outside this engine
    }
    u8 const fhd = src[4];
    u32 const dictIDCode = fhd & 3, singleSegment = (fhd >> 5) & 1, fcsID = fhd >> 6;
    u32 const didSize = dictIDCode == 3 ? 4 : dictIDCode;
    u32 const fcsSize = fcsID == 0 ? singleSegment : (1u << fcsID);
    size_t const fhs = 5 + !singleSegment + didSize + fcsSize;
    if (srcSize < fhs) return ERRV;
    if (fhd & 0x08) return ERRV;
    size_t pos = 5;
    if (!singleSegment) { u8 const wl = src[pos++]; if ((u32)(wl >> 3) + ZSA_WLOG_ABSMIN > ZSA_WLOG_MAX) return ERRV; }
    pos += didSize;
    switch (fcsID) {
    case 0: return singleSegment ? src[pos] : UNKNOWN;
    case 1: { u16 v; memcpy(&v, src + pos, 2); return (unsigned long long)v + 256; }
    case 2: { u32 v; memcpy(&v, src + pos, 4); return v; }
    default: { u64 v; memcpy(&v, src + pos, 8); return v; }
    }
}

// address of the allocation-free entry point's workspace on device `dev` (section 4 of the header), resolved when a batch or a
// decompression dictionary is created on the device, so that the entry point itself makes no runtime query
static u8* zsa_na_address(int dev);

// ---- dictionaries (device-resident digests)
#include "zsa_dict.inc"

// ---- batch object
extern "C" void zsa_batch_free(zsa_batch* b)
{
    if (!b) return;
    if (b->h_frames) (void)hipHostFree(b->h_frames);
    if (b->h_results) (void)hipHostFree(b->h_results);
    if (b->d_frames) (void)hipFree(b->d_frames);
    if (b->d_results) (void)hipFree(b->d_results);
    if (b->d_counter) (void)hipFree(b->d_counter);
    if (b->d_scratch) (void)hipFree(b->d_scratch);
    if (b->d_encScratch) (void)hipFree(b->d_encScratch);
    if (b->d_prof) (void)hipFree(b->d_prof);
    free(b);
}
#ifndef ZSA_MAX_WG_PER_CU
#define ZSA_MAX_WG_PER_CU 16
#endif
extern "C" zsa_batch* zsa_batch_create(size_t maxFrames)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); t_engineFailure = ZSA_EF_NO_DEVICE; fprintf(stderr, "libzstd_amd: no usable HIP device\n"); return NULL; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NULL;
    zsa_batch* b = (zsa_batch*)calloc(1, sizeof(zsa_batch));
    if (!b) return NULL;
    if (maxFrames == 0) maxFrames = 1;
    b->maxFrames = maxFrames;
    b->numCU = prop.multiProcessorCount;
    size_t const maxWG = (size_t)b->numCU * ZSA_MAX_WG_PER_CU;
    size_t const wgs = maxFrames < maxWG ? maxFrames : maxWG;
    b->wgs = wgs;
    b->device = dev;
    b->scratchBytes = wgs * ZSA_DEC_SCRATCH_BYTES;
    b->encScratchBytes = 0;                              // the compression workspace is sized and allocated by the compress launches
    bool ok = hipHostMalloc((void**)&b->h_frames, maxFrames * sizeof(ZsaFrameDesc), hipHostMallocDefault) == hipSuccess
           && hipHostMalloc((void**)&b->h_results, maxFrames * sizeof(u64), hipHostMallocDefault) == hipSuccess
           && hipMalloc((void**)&b->d_frames, maxFrames * sizeof(ZsaFrameDesc)) == hipSuccess
           && hipMalloc((void**)&b->d_results, maxFrames * sizeof(u64)) == hipSuccess
           && hipMalloc((void**)&b->d_counter, 256) == hipSuccess
           && hipMalloc((void**)&b->d_prof, ZSA_NPROF * sizeof(u64)) == hipSuccess
           && hipMalloc((void**)&b->d_scratch, b->scratchBytes) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); t_engineFailure = ZSA_EF_MEMORY; fprintf(stderr, "libzstd_amd: device allocation failed (%zu frames)\n", maxFrames); zsa_batch_free(b); return NULL; }
    (void)zsa_na_address(dev);
    return b;
}
extern "C" size_t zsa_batch_set_frame(zsa_batch* b, size_t i, const void* d_src, size_t srcSize, void* d_dst, size_t dstCapacity)
{
    if (!b || i >= b->maxFrames) return (size_t)ZSA_ERR(ZSA_parameter_outOfBound);
    b->h_frames[i].src = (const u8*)d_src; b->h_frames[i].dst = (u8*)d_dst;
    b->h_frames[i].srcSize = srcSize; b->h_frames[i].dstCapacity = dstCapacity;
    return 0;
}
extern "C" size_t zsa_batch_set_frames_strided(zsa_batch* b, size_t nFrames, const void* d_src, size_t srcStride, const size_t* srcSizes,
                                               void* d_dst, size_t dstStride, size_t dstCapacity)
{
    if (!b || nFrames > b->maxFrames) return (size_t)ZSA_ERR(ZSA_parameter_outOfBound);
    for (size_t i = 0; i < nFrames; i++) {
        b->h_frames[i].src = (const u8*)d_src + i * srcStride; b->h_frames[i].dst = (u8*)d_dst + i * dstStride;
        b->h_frames[i].srcSize = srcSizes ? srcSizes[i] : srcStride; b->h_frames[i].dstCapacity = dstCapacity;
    }
    return 0;
}
extern "C" size_t zsa_batch_link_src_sizes(zsa_batch* consumer, const zsa_batch* producer, void* stream)
{
    if (!consumer || !producer || consumer->nFrames > producer->nFrames) return (size_t)ZSA_ERR(ZSA_parameter_outOfBound);
    if (consumer->nFrames == 0) return 0;
    u32 const n = (u32)consumer->nFrames;
    hipLaunchKernelGGL(zsa_k_link_sizes, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, consumer->d_frames, producer->d_results, n);
    HIPCHK(hipGetLastError());
    consumer->sizesOnDevice = true; consumer->planValid = false;
    return 0;
}
extern "C" size_t zsa_batch_upload(zsa_batch* b, size_t nFrames, void* stream)
{
    if (!b || nFrames > b->maxFrames) return (size_t)ZSA_ERR(ZSA_parameter_outOfBound);
    b->nFrames = nFrames;
    b->sizesOnDevice = false; b->planValid = false;
    u64 m = 0;
    for (size_t i = 0; i < nFrames; i++) if (b->h_frames[i].srcSize > m) m = b->h_frames[i].srcSize;
    b->maxSrcSize = m;
    if (nFrames) HIPCHK(hipMemcpyAsync(b->d_frames, b->h_frames, nFrames * sizeof(ZsaFrameDesc), hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}
static unsigned zsa_grid(const zsa_batch* b, size_t ldsPerWG)
{
    size_t perCU = (160 * 1024) / (ldsPerWG ? ldsPerWG : 1);
    if (perCU > ZSA_MAX_WG_PER_CU) perCU = ZSA_MAX_WG_PER_CU;
    if (const char* e = getenv("ZSA_WG_PER_CU")) { size_t const v = (size_t)atoi(e); if (v >= 1 && v < perCU) perCU = v; }   // tuning knob
    if (perCU < 1) perCU = 1;
    size_t g = (size_t)b->numCU * perCU;
    if (g > b->nFrames) g = b->nFrames;
    return (unsigned)(g ? g : 1);
}
// A batch, a dictionary digest and the buffers its frames point to live on one device; a host thread that has switched
// devices in between (multi-GPU process, a Go thread that migrated) would hand one GPU's pointers to another's kernel.
static size_t zsa_check_device(const zsa_batch* b, int dictDevice)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "libzstd_amd: no usable HIP device -- this library has no CPU path\n"); return zsa_fail(ZSA_EF_NO_DEVICE, kErrGeneric); }
    if (dev != b->device || (dictDevice >= 0 && dictDevice != b->device)) {
        fprintf(stderr, "libzstd_amd: device mismatch: current HIP device %d, batch created on %d%s -- refusing to launch\n",
                dev, b->device, (dictDevice >= 0 && dictDevice != b->device) ? ", dictionary on another device" : "");
        return zsa_fail(ZSA_EF_DEVICE_MISMATCH, (size_t)ZSA_ERR(ZSA_stage_wrong));
    }
    return 0;
}

// What a compress launch of this batch needs: the match-table bytes of its most demanding frame (sizes the workspace).
static void zsa_plan_compress(zsa_batch* b, int level, const zsa_CDict* cdict)
{
    if (b->planValid && b->planLevel == level && b->planDict == (const void*)cdict) return;
    b->planLevel = level; b->planDict = cdict; b->planValid = true; b->planHtBytes = 4096;
    if (b->sizesOnDevice) { b->planHtBytes = ZSA_HT_BYTES_UNSIZED; return; }   // sizes unknown here: a fixed arena (larger tables are refused per frame)
    for (size_t i = 0; i < b->nFrames; i++) {
        ZsaCParams cp; int dictMode = zsa_dm_none;
        u64 const n = b->h_frames[i].srcSize;
        int const bad = cdict ? zsa_resolveCDictUse(&cp, &dictMode, cdict->cp, cdict->level, cdict->dictSize, n)
                              : zsa_resolveCParams(&cp, level, n, 0, zsa_cpm_noAttachDict);
        if (bad) continue;
        u64 const t = ((u64)4 << cp.hlog) + (cdict && cp.strat >= 2 ? ((u64)4 << (cp.hlog > cp.clog ? cp.hlog : cp.clog))
                                                                    : zsa_auxTableBytes(cp.strat, cp.wlog, cp.hlog, cp.clog));
        if (t <= ZSA_HT_BYTES && t > b->planHtBytes) b->planHtBytes = t;   // (larger ones are refused per frame by the kernel)
    }
}
extern "C" size_t zsa_compress_enqueue(zsa_batch* b, int level, const zsa_CDict* cdict, void* stream)
{
    t_engineFailure = ZSA_EF_NONE;
    if (!b) return kErrGeneric;
    if (b->nFrames == 0) return 0;
    { size_t const e = zsa_check_device(b, cdict ? cdict->device : -1); if (e) return e; }
    zsa_plan_compress(b, level, cdict);
    ZsaEncLayout const L = zsa_enc_layout(b->sizesOnDevice ? ZSA_BLOCKSIZE_MAX : b->maxSrcSize, b->planHtBytes);
    size_t const ldsPerWG = sizeof(ZsaEncShared) + 64;
    unsigned grid = zsa_grid(b, ldsPerWG);
    if ((size_t)grid * L.stride > b->encScratchBytes) {
        // the workspace of a full grid may not fit what the device has free (large tables of high levels, a GPU that is partly
        // in use): the launch then keeps fewer frames in flight -- at most half of what is free plus what the batch holds
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); freeB = 0; }
        size_t const budget = (freeB + b->encScratchBytes) / 2;
        if ((size_t)grid * L.stride > budget) { size_t const g = budget / L.stride; grid = (unsigned)(g ? g : 1); }
    }
    {   // workspace: grows when a launch needs more than the batch has held so far; steady-state enqueues allocate nothing
        size_t need = (size_t)grid * L.stride;
        if (need > b->encScratchBytes) {
            if (b->d_encScratch) { (void)hipFree(b->d_encScratch); b->d_encScratch = NULL; b->encScratchBytes = 0; }   // (waits for launches using it)
            bool got = false;
            for (int attempt = 0; attempt < 2 && !got; attempt++) {   // a refused allocation is tried once more with half the frames in flight
                if (attempt) { if (grid < 2) break; grid = (grid + 1) / 2; need = (size_t)grid * L.stride; }
                got = !zsa_forced_alloc_failure() && hipMalloc((void**)&b->d_encScratch, need) == hipSuccess;
                if (!got) (void)hipGetLastError();
            }
            if (!got) {
                fprintf(stderr, "libzstd_amd: device allocation of the compression workspace failed (%zu bytes)\n", need);
                b->d_encScratch = NULL;
                return zsa_fail(ZSA_EF_MEMORY, (size_t)ZSA_ERR(ZSA_memory_allocation));
            }
            b->encScratchBytes = need;
        }
    }
    HIPCHK(hipMemsetAsync(b->d_counter, 0, 256, (hipStream_t)stream));
    // which parser families the level's clevels.h rows can resolve to: a dictionary -> family 0; ZSTD_fast at levels <= 2
    // (level 0 means 3) -> family 2; ZSTD_dfast at levels 2-4 -> family 3; ZSTD_greedy/ZSTD_lazy/ZSTD_lazy2 from level 4 on -> family 1
    int const lv = level == 0 ? 3 : level;
    // (a CDict's strategy is fixed at its creation; frames that re-resolve their parameters -- 128 KiB and more -- may fall to ZSTD_fast)
    bool const fam0 = cdict != nullptr, fam4 = cdict != nullptr && cdict->cp.strat >= 2;
    bool const fam2 = !cdict && lv <= 2, fam3 = !cdict && lv >= 2 && lv <= 4, fam1 = !cdict && lv >= 4;
    int const famMask = (fam0 ? 1 : 0) | (fam1 ? 2 : 0) | (fam2 ? 4 : 0) | (fam3 ? 8 : 0) | (fam4 ? 16 : 0);
    const ZsaCDictDev* const cd = cdict ? (const ZsaCDictDev*)cdict->d_cd : (const ZsaCDictDev*)nullptr;
    u64* const prof = b->profOn ? b->d_prof : (u64*)nullptr;
    // (same scratch: the launches run one after the other on the stream; the first one reports unsupported frames)
    int reporter = 1;
    if (fam2) {
        hipLaunchKernelGGL(zsa_k_compress, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           b->d_frames, b->d_results, (u32)b->nFrames, b->d_counter + 32, b->d_encScratch, L, level, famMask, reporter, prof);
        reporter = 0;
    }
    if (fam0) {
        hipLaunchKernelGGL(zsa_k_compress_dict, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           b->d_frames, b->d_results, (u32)b->nFrames, b->d_counter, b->d_encScratch, L, level, cd, famMask, reporter, prof);
        reporter = 0;
    }
    if (fam4) {
        hipLaunchKernelGGL(zsa_k_compress_dict_hi, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           b->d_frames, b->d_results, (u32)b->nFrames, b->d_counter + 8, b->d_encScratch, L, level, cd, famMask, reporter, prof);
        reporter = 0;
    }
    if (fam3) {
        hipLaunchKernelGGL(zsa_k_compress_dfast, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           b->d_frames, b->d_results, (u32)b->nFrames, b->d_counter + 48, b->d_encScratch, L, level, famMask, reporter, prof);
        reporter = 0;
    }
    if (fam1)
        hipLaunchKernelGGL(zsa_k_compress_lazy, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           b->d_frames, b->d_results, (u32)b->nFrames, b->d_counter + 16, b->d_encScratch, L, level, famMask, reporter, prof);
    HIPCHK(hipGetLastError());
    b->lastWG = grid; b->lastLDS = (unsigned)ldsPerWG;
    return 0;
}
extern "C" size_t zsa_decompress_enqueue(zsa_batch* b, const zsa_DDict* ddict, void* stream)
{
    t_engineFailure = ZSA_EF_NONE;
    if (!b) return kErrGeneric;
    if (b->nFrames == 0) return 0;
    { size_t const e = zsa_check_device(b, ddict ? ddict->device : -1); if (e) return e; }
    size_t const ldsPerWG = sizeof(ZsaDecShared) + 64;
    unsigned const grid = zsa_grid(b, ldsPerWG);
    HIPCHK(hipMemsetAsync(b->d_counter, 0, 256, (hipStream_t)stream));
    hipLaunchKernelGGL(zsa_k_decompress, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                       b->d_frames, b->d_results, (u32)b->nFrames, b->d_counter, b->d_scratch,
                       ddict ? (const ZsaDDictDev*)ddict->d_dd : (const ZsaDDictDev*)nullptr, b->profOn ? b->d_prof : (u64*)nullptr);
    HIPCHK(hipGetLastError());
    b->lastWG = grid; b->lastLDS = (unsigned)ldsPerWG;
    return 0;
}
extern "C" size_t zsa_batch_results(zsa_batch* b, size_t* outSize, void* stream)
{
    if (!b) return kErrGeneric;
    if (b->nFrames) HIPCHK(hipMemcpyAsync(b->h_results, b->d_results, b->nFrames * sizeof(u64), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (size_t i = 0; i < b->nFrames; i++) outSize[i] = (size_t)b->h_results[i];
    return 0;
}
extern "C" size_t zsa_batch_profile(zsa_batch* b, int enable, uint64_t* out /* ZSA_NPROF words or NULL */, void* stream)
{   // diagnostics: per-phase shader-clock cycles of lane 0 summed over all workgroups since profiling was enabled
    if (!b) return kErrGeneric;
    if (out) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); HIPCHK(hipMemcpy(out, b->d_prof, ZSA_NPROF * sizeof(u64), hipMemcpyDeviceToHost)); }
    if (enable && !b->profOn) HIPCHK(hipMemset(b->d_prof, 0, ZSA_NPROF * sizeof(u64)));
    b->profOn = enable;
    return 0;
}
extern "C" const uint64_t* zsa_batch_device_results(const zsa_batch* b) { return b ? b->d_results : NULL; }
extern "C" size_t zsa_batch_export_sizes_i32(const zsa_batch* b, int32_t* d_out, void* stream)
{
    if (!b || !d_out) return kErrGeneric;
    if (b->nFrames == 0) return 0;
    u32 const n = (u32)b->nFrames;
    hipLaunchKernelGGL(zsa_k_export_sizes, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->d_results, d_out, n);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" void zsa_batch_last_launch(const zsa_batch* b, unsigned* wg, unsigned* lds) { if (b) { if (wg) *wg = b->lastWG; if (lds) *lds = b->lastLDS; } }

// ---- one entry point that allocates nothing (include/zstd_amd.h section 4): the decompress kernel's workspace -- the work
// counter and, per workgroup, the literal buffer and the parked tables -- lives in a file-scope device array instead of
// hipMalloc'ed memory.  Each GPU holds its own copy of the array; calls on one GPU share it, so overlapping calls race.
#define ZSA_NA_WORKGROUPS 128u
#define ZSA_NA_MAX_FRAMES ((size_t)1 << 20)
static __device__ __align__(256) u8 zsa_na_scratch[256 + (size_t)ZSA_NA_WORKGROUPS * ZSA_DEC_SCRATCH_BYTES];
#define ZSA_NA_MAX_DEVICES 64
static u8* g_naPtr[ZSA_NA_MAX_DEVICES];
static std::mutex g_naMutex;
static u8* zsa_na_address(int dev)
{
    if (dev < 0 || dev >= ZSA_NA_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> g(g_naMutex);
    if (!g_naPtr[dev]) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != dev) { (void)hipGetLastError(); return nullptr; }   // (the lookup answers for the current device)
        void* p = nullptr;
        hipError_t const e = hipGetSymbolAddress(&p, HIP_SYMBOL(zsa_na_scratch));
        if (e != hipSuccess) { fprintf(stderr, "libzstd_amd: hipGetSymbolAddress(zsa_na_scratch) failed: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); return nullptr; }
        g_naPtr[dev] = (u8*)p;
    }
    return g_naPtr[dev];
}
// warm-up for callers that make their first call under graph capture: resolves the workspace address on the current device
extern "C" size_t zsa_noalloc_prepare(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return zsa_fail(ZSA_EF_NO_DEVICE, kErrGeneric); }
    return zsa_na_address(dev) ? 0 : zsa_fail(ZSA_EF_RUNTIME, kErrGeneric);
}
extern "C" size_t zsa_decompress_frames_noalloc(const zsa_frame* d_frames, uint64_t* d_results, size_t nFrames, const zsa_DDict* ddict, void* stream)
{
    static_assert(sizeof(zsa_frame) == sizeof(ZsaFrameDesc), "zsa_frame is the kernels' frame descriptor");
    t_engineFailure = ZSA_EF_NONE;
    if (nFrames == 0) return 0;
    if (!d_frames || !d_results) return kErrGeneric;
    if (nFrames > ZSA_NA_MAX_FRAMES) {
        fprintf(stderr, "libzstd_amd: zsa_decompress_frames_noalloc: %zu frames exceed the limit of %zu per call\n", nFrames, ZSA_NA_MAX_FRAMES);
        return (size_t)ZSA_ERR(ZSA_parameter_outOfBound);
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ZSA_NA_MAX_DEVICES) { (void)hipGetLastError(); fprintf(stderr, "libzstd_amd: no usable HIP device -- this library has no CPU path\n"); return zsa_fail(ZSA_EF_NO_DEVICE, kErrGeneric); }
    if (ddict && ddict->device != dev) { fprintf(stderr, "libzstd_amd: device mismatch: dictionary on device %d, current device %d\n", ddict->device, dev); return zsa_fail(ZSA_EF_DEVICE_MISMATCH, (size_t)ZSA_ERR(ZSA_stage_wrong)); }
    // the array's address on THIS device (the name itself is only a host-side handle): looked up once per device, by
    // zsa_batch_create / zsa_createDDict / zsa_noalloc_prepare, or here on a first call that none of them preceded
    u8* const ptr = zsa_na_address(dev);
    if (!ptr) return zsa_fail(ZSA_EF_RUNTIME, kErrGeneric);
    unsigned const grid = (unsigned)(nFrames < ZSA_NA_WORKGROUPS ? nFrames : ZSA_NA_WORKGROUPS);
    HIPCHK(hipMemsetAsync(ptr, 0, 256, (hipStream_t)stream));   // the work counter starts at zero; the rest is write-before-read
    hipLaunchKernelGGL(zsa_k_decompress, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                       (const ZsaFrameDesc*)d_frames, (u64*)d_results, (u32)nFrames, (u32*)ptr, ptr + 256,
                       ddict ? (const ZsaDDictDev*)ddict->d_dd : (const ZsaDDictDev*)nullptr, (u64*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- host-memory batches: stage through pinned buffers, run the device batch, copy results back
// One per calling thread (the reference's call sites are lock-free and may run on any goroutine/OS thread, zstd.go:95-129):
// sized to the requests the thread has made, released when the thread ends.  At process exit the HIP runtime may already be
// gone when thread-local destructors run, so from then on everything is left to it.
static bool g_zsaExiting = false;
static void zsa_mark_exit() { g_zsaExiting = true; }
#define ZSA_MAX_SUBS 8          // a large host batch runs as up to this many sub-batches, each on a stream of its own
struct ZsaStage {
    u8 *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    u64 *h_meta = nullptr, *d_meta = nullptr;   // per frame: pack records (large batches only)
    size_t inCap = 0, outCap = 0, metaCap = 0;
    zsa_batch* batch[ZSA_MAX_SUBS] = {};
    hipStream_t stream[ZSA_MAX_SUBS] = {};
    hipEvent_t uploaded[ZSA_MAX_SUBS] = {};
    int device = -1;
    void forget()
    {
        h_in = h_out = d_in = d_out = nullptr; h_meta = d_meta = nullptr; inCap = outCap = metaCap = 0;
        for (int k = 0; k < ZSA_MAX_SUBS; k++) { batch[k] = nullptr; stream[k] = nullptr; uploaded[k] = nullptr; }
    }
    void release()
    {
        if (h_in) { (void)hipHostFree(h_in); (void)hipFree(d_in); }
        if (h_out) { (void)hipHostFree(h_out); (void)hipFree(d_out); }
        if (h_meta) { (void)hipHostFree(h_meta); (void)hipFree(d_meta); }
        for (int k = 0; k < ZSA_MAX_SUBS; k++) { if (batch[k]) zsa_batch_free(batch[k]); if (stream[k]) (void)hipStreamDestroy(stream[k]); if (uploaded[k]) (void)hipEventDestroy(uploaded[k]); }
        forget();
    }
    ~ZsaStage()
    {
        if (g_zsaExiting || device == -1) return;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return;
        if (cur != device && hipSetDevice(device) != hipSuccess) return;
        release();
        if (cur != device) (void)hipSetDevice(cur);
    }
};
static thread_local ZsaStage t_stage;
// Lets a long-lived thread hand its staging buffers back without ending (no counterpart in libzstd, whose contexts are
// freed by ZSTD_freeCCtx/ZSTD_freeDCtx: zstd_ctx.go:43-51).
extern "C" void zsa_thread_release(void) { if (t_stage.device != -1) { t_stage.release(); t_stage.device = -1; } }

// ---- pinned host memory handed out by the library: buffers a caller takes from zsa_host_alloc are known to be page-locked
// and mapped into every GPU's address space, so a host batch whose payloads (or destinations) all lie inside them skips the
// staging copy on that side and the GPU moves the bytes itself.  (The ranges are kept here: asking the runtime about
// every pointer of a batch would cost more than the copy it saves.)
static std::mutex g_pinMutex;
static std::map<uintptr_t, size_t> g_pinned;     // base -> bytes
extern "C" void* zsa_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0) bytes = 1;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); t_engineFailure = ZSA_EF_MEMORY; return nullptr; }
    std::lock_guard<std::mutex> g(g_pinMutex);
    g_pinned[(uintptr_t)p] = bytes;
    return p;
}
extern "C" void zsa_host_free(void* p)
{
    if (!p) return;
    {   std::lock_guard<std::mutex> g(g_pinMutex);
        g_pinned.erase((uintptr_t)p);
    }
    (void)hipHostFree(p);
}
// every non-empty range [ptr[i], ptr[i] + len[i]) lies inside memory from zsa_host_alloc
static bool zsa_all_pinned(const void* const* ptr, const size_t* len, size_t n)
{
    std::lock_guard<std::mutex> g(g_pinMutex);
    if (g_pinned.empty()) return false;
    uintptr_t lo = 0, hi = 0;                    // the range that held the previous pointer (batches come from one slab as a rule)
    for (size_t i = 0; i < n; i++) {
        if (len[i] == 0) continue;
        uintptr_t const a = (uintptr_t)ptr[i];
        if (!(a >= lo && a + len[i] <= hi)) {
            auto it = g_pinned.upper_bound(a);
            if (it == g_pinned.begin()) return false;
            --it;
            lo = it->first; hi = lo + it->second;
            if (!(a >= lo && a + len[i] <= hi)) return false;
        }
    }
    return true;
}

static size_t zsa_stage_reserve(ZsaStage& st, size_t inBytes, size_t outBytes, const size_t* subFrames, unsigned nSubs, size_t metaBytes, bool pinIn, bool pinOut)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "libzstd_amd: no usable HIP device -- this library has no CPU path\n"); return zsa_fail(ZSA_EF_NO_DEVICE, kErrGeneric); }
    if (st.device != dev) {   // first use on this thread, or the thread moved to another device: staging is per device
        if (st.device != -1) {   // (what it holds belongs to the old device)
            int const old = st.device;
            if (hipSetDevice(old) == hipSuccess) { st.release(); (void)hipSetDevice(dev); }
            else st.forget();
        }
        static std::once_flag once;
        std::call_once(once, [] { atexit(zsa_mark_exit); });
        st.device = dev;
    }
    for (unsigned k = 0; k < nSubs; k++) if (!st.stream[k]) { HIPCHK(hipStreamCreateWithFlags(&st.stream[k], hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&st.uploaded[k], hipEventDisableTiming)); }
    bool const forced = zsa_forced_alloc_failure();
    // (a side whose bytes the GPU moves itself needs no pinned staging copy, only the device buffer)
    auto const grow = [&](u8*& h, u8*& d, size_t& cap, size_t bytes, bool needHost) -> bool {
        if (bytes <= cap && (!needHost || h)) return true;
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr; cap = 0;
        size_t const want = bytes + (bytes >> 2) + 4096;
        if (forced || hipMalloc((void**)&d, want) != hipSuccess) { (void)hipGetLastError(); d = nullptr; return false; }
        if (needHost && hipHostMalloc((void**)&h, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); h = d = nullptr; return false; }
        cap = want;
        return true;
    };
    if (!grow(st.h_in, st.d_in, st.inCap, inBytes, !pinIn)) return zsa_fail(ZSA_EF_MEMORY, kErrMemory);
    if (!grow(st.h_out, st.d_out, st.outCap, outBytes, !pinOut)) return zsa_fail(ZSA_EF_MEMORY, kErrMemory);
    if (metaBytes > st.metaCap) {
        if (st.h_meta) { (void)hipHostFree(st.h_meta); (void)hipFree(st.d_meta); st.h_meta = st.d_meta = nullptr; st.metaCap = 0; }
        size_t const cap = metaBytes + (metaBytes >> 2) + 4096;
        if (forced || hipHostMalloc((void**)&st.h_meta, cap, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&st.d_meta, cap) != hipSuccess) { (void)hipGetLastError(); return zsa_fail(ZSA_EF_MEMORY, kErrMemory); }
        st.metaCap = cap;
    }
    for (unsigned k = 0; k < nSubs; k++) {
        if (!st.batch[k] || st.batch[k]->maxFrames < subFrames[k]) {
            if (st.batch[k]) zsa_batch_free(st.batch[k]);
            size_t const cap = subFrames[k] + (subFrames[k] >> 2);   // sized to the request, geometric growth: a thread that compresses one
            st.batch[k] = forced ? nullptr : zsa_batch_create(cap);   // payload at a time holds one workgroup's workspace, not a chip's
            if (!st.batch[k]) return zsa_fail(ZSA_EF_MEMORY, kErrMemory);
        }
    }
    return 0;
}

// ---- large host batches.  What a host batch spends its time on is moving bytes: caller's buffers -> pinned staging -> device,
// and back.  So (1) from ZSA_PAR_MIN_BYTES on the copies between the caller's (pageable) buffers and the pinned staging are
// made by a few helper threads, piece by piece, while the pieces already staged cross PCIe; (2) the batch is cut into up to
// ZSA_MAX_SUBS sub-batches, each with its own zsa_batch and stream, so that the upload of sub-batch k + 1, the kernel of
// sub-batch k and the download of sub-batch k - 1 run at the same time; (3) compressed output is packed on the device first,
// so that only the bytes produced travel back; (4) payloads / decompression destinations in memory from zsa_host_alloc cross
// PCIe straight from / into the caller's buffers, with no staging copy on that side.  Helper threads make no HIP calls.
// ZSTD_AMD_COPY_THREADS (default 8, 1 = none) sets their number.
#define ZSA_PAR_MIN_BYTES ((size_t)8 << 20)
#define ZSA_SUB_MIN_BYTES ((size_t)24 << 20)     // a sub-batch is worth its launch from about this many staged bytes on
static unsigned zsa_copy_threads()
{
    static unsigned const conf = [] { const char* e = getenv("ZSTD_AMD_COPY_THREADS"); int const v = e ? atoi(e) : 8; return (unsigned)(v < 1 ? 1 : (v > 16 ? 16 : v)); }();
    return conf;
}
static unsigned zsa_max_subs()
{
    static unsigned const conf = [] { const char* e = getenv("ZSTD_AMD_SUBS"); int const v = e ? atoi(e) : 4; return (unsigned)(v < 1 ? 1 : (v > ZSA_MAX_SUBS ? ZSA_MAX_SUBS : v)); }();
    return conf;
}
// frames [lo, hi) cut into pieces of about equal staged size: piece p = frames [cut[p], cut[p + 1])
static std::vector<size_t> zsa_cut_pieces(const std::vector<size_t>& off, size_t lo, size_t hi, size_t pieces)
{
    std::vector<size_t> cut; cut.push_back(lo);
    size_t const total = off[hi] - off[lo];
    for (size_t p = 1; p < pieces; p++) {
        size_t const want = off[lo] + total / pieces * p;
        size_t f = cut.back();
        while (f < hi && off[f] < want) f++;
        if (f > cut.back() && f < hi) cut.push_back(f);
    }
    cut.push_back(hi);
    return cut;
}
struct ZsaPieceFlags {
    std::vector<std::atomic<int>> f;
    explicit ZsaPieceFlags(size_t n) : f(n) { for (auto& x : f) x.store(0, std::memory_order_relaxed); }
    void set(size_t p, int v = 1) { f[p].store(v, std::memory_order_release); }
    int wait(size_t p) { unsigned spins = 0; int v; while (!(v = f[p].load(std::memory_order_acquire))) { if (++spins > 64) std::this_thread::yield(); } return v; }
};

// packs the produced bytes of every frame back to back (16-byte aligned offsets): one workgroup per frame
__global__ void __launch_bounds__(256) zsa_k_pack(const u8* __restrict__ from, u8* __restrict__ to, const u64* __restrict__ meta, u32 n)
{
    u32 const i = blockIdx.x;
    if (i >= n) return;
    u64 const srcOff = meta[3 * (size_t)i], dstOff = meta[3 * (size_t)i + 1], size = meta[3 * (size_t)i + 2];
    const uint4* const a = (const uint4*)(from + srcOff);
    uint4* const b = (uint4*)(to + dstOff);
    u64 const q = size >> 4;
    for (u64 k = threadIdx.x; k < q; k += 256) b[k] = a[k];
    for (u64 k = (q << 4) + threadIdx.x; k < size; k += 256) to[dstOff + k] = from[srcOff + k];
}
static bool zsa_trace_on() { static int const on = getenv("ZSTD_AMD_TRACE") ? 1 : 0; return on != 0; }
#define ZSA_TRACE(...) do { if (zsa_trace_on()) { fprintf(stderr, "[zsa %8.3f ms] ", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
static size_t zsa_run_host_batch(bool compress, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity,
                                 size_t* outSize, size_t n, int level, const zsa_CDict* cd, const zsa_DDict* dd)
{
    auto const t0_ = std::chrono::steady_clock::now();
    t_engineFailure = ZSA_EF_NONE;
    if (n == 0) return 0;
    ZsaStage& st = t_stage;
    // Payloads / destinations in memory from zsa_host_alloc cross PCIe straight from / into the caller's buffers, one DMA
    // copy per RUN of frames that lie one behind the other there: the device slots of a run mirror its layout on the host.
    // (Compressed output is sparse in its bound-sized slots, so compress destinations always go through the packed staging.)
    bool pinIn = zsa_all_pinned(src, srcSize, n), pinOut = !compress && zsa_all_pinned((const void* const*)dst, dstCapacity, n);
    // layout: inputs and outputs back to back, 16-byte aligned (pinned side: congruent to the host address modulo 16), with 32
    // bytes of slack after each input.  A compress slot holds ZSTD_compressBound(srcSize) bytes whatever the caller's capacity: as
    // ZSTD_compress does, a destination below the bound is served when the frame fits it (zstd_compress.c:5090-5111 has no
    // bound check; the kernels keep the bound as their contract)
    std::vector<size_t> inOff(n + 1), outOff(n + 1), devCap(n);
    std::vector<char> inRun(n, 0), outRun(n, 0);            // frame i continues the run of frame i - 1
    size_t inBytes = 0, outBytes = 0;
    auto const layout = [&]() {
        size_t runsIn = 0, runsOut = 0;
        inBytes = outBytes = 0;
        for (size_t i = 0; i < n; i++) {
            inRun[i] = outRun[i] = 0;
            if (pinIn) {
                uintptr_t const p = (uintptr_t)src[i], prev = i ? (uintptr_t)src[i - 1] : 0;
                if (i && srcSize[i - 1] && srcSize[i] && p >= prev + srcSize[i - 1] && p - (prev + srcSize[i - 1]) <= 4096) { inRun[i] = 1; inOff[i] = inOff[i - 1] + (p - prev); }
                else { inOff[i] = ((inBytes + 15) & ~(size_t)15) + (p & 15); runsIn++; }
                if (inOff[i] + srcSize[i] + 48 > inBytes) inBytes = inOff[i] + srcSize[i] + 48;
            } else { inOff[i] = inBytes; inBytes += (srcSize[i] + 32 + 15) & ~(size_t)15; }
            size_t c = dstCapacity[i];
            if (compress) { size_t const bnd = zsa_compressBound(srcSize[i]); if (!zsa_isError(bnd) && c < bnd) c = bnd; }
            devCap[i] = c;
            if (pinOut) {   // (a run is copied back whole, so it may hold nothing but the frames' own destinations: exactly adjacent)
                uintptr_t const p = (uintptr_t)dst[i], prev = i ? (uintptr_t)dst[i - 1] : 0;
                if (i && dstCapacity[i - 1] && c && p == prev + dstCapacity[i - 1]) { outRun[i] = 1; outOff[i] = outOff[i - 1] + dstCapacity[i - 1]; }
                else { outOff[i] = ((outBytes + 15) & ~(size_t)15) + (p & 15); runsOut++; }
                outBytes = outOff[i] + c;
            } else { outOff[i] = outBytes; outBytes += (c + 15) & ~(size_t)15; }
        }
        inBytes = (inBytes + 15) & ~(size_t)15; outBytes = (outBytes + 15) & ~(size_t)15;
        inOff[n] = inBytes; outOff[n] = outBytes;
        // scattered buffers (compressed frames in bound-sized slots, say) would mean one small copy per frame: staged instead
        bool const manyIn = pinIn && runsIn > n / 16 + 16, manyOut = pinOut && runsOut > n / 16 + 16;
        if (manyIn) pinIn = false;
        if (manyOut) pinOut = false;
        return manyIn || manyOut;
    };
    if (layout()) (void)layout();
    bool const large = n >= 8 && inBytes + outBytes >= ZSA_PAR_MIN_BYTES;
    unsigned const T = large ? zsa_copy_threads() : 1;
    // sub-batches: frames [sub[k], sub[k + 1]), about equal in staged bytes
    std::vector<size_t> sub;
    {   size_t S = large ? (inBytes + outBytes) / ZSA_SUB_MIN_BYTES : 1;
        if (S > zsa_max_subs()) S = zsa_max_subs();
        if (S > n / 8) S = n / 8;
        if (S < 1) S = 1;
        std::vector<size_t> both(n + 1);
        for (size_t i = 0; i <= n; i++) both[i] = inOff[i] + outOff[i];
        sub = zsa_cut_pieces(both, 0, n, S);
    }
    unsigned const S = (unsigned)sub.size() - 1;
    size_t subFrames[ZSA_MAX_SUBS] = {};
    for (unsigned k = 0; k < S; k++) subFrames[k] = sub[k + 1] - sub[k];
    if (!large) pinIn = pinOut = false;                      // (small calls: one staging copy each way, one wait)
    { size_t const e = zsa_stage_reserve(st, inBytes, outBytes, subFrames, S, large ? 3 * n * sizeof(u64) : 0, pinIn, pinOut); if (e) return e; }
    for (unsigned k = 0; k < S; k++)
        for (size_t i = sub[k]; i < sub[k + 1]; i++) zsa_batch_set_frame(st.batch[k], i - sub[k], st.d_in + inOff[i], srcSize[i], st.d_out + outOff[i], devCap[i]);
    auto const finish = [&](size_t i) {   // a frame that does not fit the caller's destination: the reference's error
        if (!zsa_isError(outSize[i]) && outSize[i] > dstCapacity[i]) outSize[i] = (size_t)ZSA_ERR(ZSA_dstSize_tooSmall);
    };
    if (!large) {
        hipStream_t const q = st.stream[0];
        for (size_t i = 0; i < n; i++) if (srcSize[i]) memcpy(st.h_in + inOff[i], src[i], srcSize[i]);
        HIPCHK(hipMemcpyAsync(st.d_in, st.h_in, inBytes, hipMemcpyHostToDevice, q));
        { size_t const e = zsa_batch_upload(st.batch[0], n, q); if (e) return e; }
        { size_t const e = compress ? zsa_compress_enqueue(st.batch[0], level, cd, q) : zsa_decompress_enqueue(st.batch[0], dd, q); if (e) return e; }
        // small calls (the one-shot entry points): the whole output region travels back with the result words, one wait for both
        bool const eager = outBytes <= ((size_t)256 << 10);
        if (eager) HIPCHK(hipMemcpyAsync(st.h_out, st.d_out, outBytes, hipMemcpyDeviceToHost, q));
        { size_t const e = zsa_batch_results(st.batch[0], outSize, q); if (e) return e; }
        for (size_t i = 0; i < n; i++) finish(i);
        if (!eager) {   // copy back only the span that was produced
            size_t lo = (size_t)-1, hi = 0;
            for (size_t i = 0; i < n; i++) if (!zsa_isError(outSize[i]) && outSize[i]) {
                if (outOff[i] < lo) lo = outOff[i];
                if (outOff[i] + outSize[i] > hi) hi = outOff[i] + outSize[i];
            }
            if (hi <= lo) return 0;
            HIPCHK(hipMemcpyAsync(st.h_out + lo, st.d_out + lo, hi - lo, hipMemcpyDeviceToHost, q));
            HIPCHK(hipStreamSynchronize(q));
        }
        for (size_t i = 0; i < n; i++) if (!zsa_isError(outSize[i]) && outSize[i]) memcpy(dst[i], st.h_out + outOff[i], outSize[i]);
        return 0;
    }

    // ---- large batch
    ZSA_TRACE("%s: %zu frames, %zu MiB in, %zu MiB out slots, %u sub-batches, %u helpers, pinned in/out %d/%d", compress ? "compress" : "decompress", n, inBytes >> 20, outBytes >> 20, S, T, (int)pinIn, (int)pinOut);
    u64* const h_pk = st.h_meta; u64* const d_pk = st.d_meta;   // 3 words per frame for zsa_k_pack
    size_t firstError = 0;
    // input side, sub-batch after sub-batch; the kernel of each is enqueued as soon as its input is on its way
    {   std::vector<std::vector<size_t>> cuts(S);
        std::vector<size_t> pieceBase(S + 1, 0);
        for (unsigned k = 0; k < S; k++) { cuts[k] = zsa_cut_pieces(inOff, sub[k], sub[k + 1], pinIn ? 1 : (size_t)T); pieceBase[k + 1] = pieceBase[k] + cuts[k].size() - 1; }
        size_t const P = pieceBase[S];
        ZsaPieceFlags staged(P);
        std::vector<std::thread> helpers;
        unsigned made = 0;
        u8* const h_in = st.h_in;
        auto const work = [&](unsigned w) {   // helper w stages pieces w, w + T, ... in order (pieces are numbered across sub-batches)
            for (size_t p = w; p < P; p += T) {
                unsigned k = 0; while (p >= pieceBase[k + 1]) k++;
                size_t const q = p - pieceBase[k];
                for (size_t i = cuts[k][q]; i < cuts[k][q + 1]; i++) if (srcSize[i]) memcpy(h_in + inOff[i], src[i], srcSize[i]);
                staged.set(p);
            }
        };
        if (!pinIn) {
            try { for (; made < T; made++) helpers.emplace_back(work, made); } catch (...) { }
            for (unsigned w = made; w < T; w++) work(w);   // (a helper that could not be started: its share is copied here)
        }
        for (unsigned k = 0; k < S && !firstError; k++) {
            hipStream_t const q = st.stream[k];
            size_t const a = sub[k], b = sub[k + 1];
            hipError_t err = hipSuccess;
            // uploads go one after the other (sub-batch k + 1's waits for sub-batch k's), so that kernel k starts as early as it can
            if (k && pinIn) err = hipStreamWaitEvent(q, st.uploaded[k - 1], 0);
            if (pinIn && err == hipSuccess) {   // one DMA copy per run, straight from the caller's memory
                for (size_t i = a; i < b && err == hipSuccess; ) {
                    size_t j = i + 1;
                    while (j < b && inRun[j]) j++;
                    size_t const bytes = ((uintptr_t)src[j - 1] + srcSize[j - 1]) - (uintptr_t)src[i];
                    if (bytes) err = hipMemcpyAsync(st.d_in + inOff[i], src[i], bytes, hipMemcpyHostToDevice, q);
                    i = j;
                }
                if (err == hipSuccess) err = hipEventRecord(st.uploaded[k], q);
            } else {
                for (size_t p = pieceBase[k]; p < pieceBase[k + 1]; p++) {
                    staged.wait(p);
                    size_t const c0 = cuts[k][p - pieceBase[k]], c1 = cuts[k][p - pieceBase[k] + 1];
                    size_t const lo = inOff[c0], hi = inOff[c1];
                    if (err == hipSuccess && hi > lo) err = hipMemcpyAsync(st.d_in + lo, st.h_in + lo, hi - lo, hipMemcpyHostToDevice, q);
                }
            }
            if (err != hipSuccess) { fprintf(stderr, "libzstd_amd: host batch upload failed: %s\n", hipGetErrorString(err)); (void)hipGetLastError(); firstError = zsa_fail(ZSA_EF_RUNTIME, kErrGeneric); break; }
            size_t e = zsa_batch_upload(st.batch[k], b - a, q);
            if (!e) e = compress ? zsa_compress_enqueue(st.batch[k], level, cd, q) : zsa_decompress_enqueue(st.batch[k], dd, q);
            if (!e && pinOut) {   // one DMA copy per run, straight into the caller's memory (a frame that failed leaves its slot's bytes there)
                for (size_t i = a; i < b && err == hipSuccess; ) {
                    size_t j = i + 1;
                    while (j < b && outRun[j]) j++;
                    size_t const bytes = ((uintptr_t)dst[j - 1] + dstCapacity[j - 1]) - (uintptr_t)dst[i];
                    if (bytes) err = hipMemcpyAsync(dst[i], st.d_out + outOff[i], bytes, hipMemcpyDeviceToHost, q);
                    i = j;
                }
                if (err != hipSuccess) { (void)hipGetLastError(); e = zsa_fail(ZSA_EF_RUNTIME, kErrGeneric); }
            }
            if (e) firstError = e;
            ZSA_TRACE("sub-batch %u enqueued", k);
        }
        for (auto& h : helpers) h.join();
        if (firstError) { for (unsigned k = 0; k < S; k++) (void)hipStreamSynchronize(st.stream[k]); return firstError; }
    }
    // output side: results of sub-batch k (this waits for its kernel), then its bytes -- while the later sub-batches run
    if (pinOut) {
        for (unsigned k = 0; k < S; k++) {
            size_t const e = zsa_batch_results(st.batch[k], outSize + sub[k], st.stream[k]);   // (synchronises the stream: the copies into the destinations are done)
            if (e && !firstError) firstError = e;
            ZSA_TRACE("sub-batch %u done", k);
        }
        for (size_t i = 0; i < n; i++) finish(i);
        return firstError;
    }
    {   std::vector<size_t> hostOff(n + 1, 0);                 // where frame i's bytes land in the pinned buffer they travel through
        std::vector<std::vector<size_t>> cuts(S);
        std::vector<size_t> pieceBase(S + 1, 0);
        std::vector<const u8*> hFrom(S, nullptr);
        // pieces are cut before the sizes are known: by capacity, which bounds what is produced
        for (unsigned k = 0; k < S; k++) { cuts[k] = zsa_cut_pieces(outOff, sub[k], sub[k + 1], (size_t)T); pieceBase[k + 1] = pieceBase[k] + cuts[k].size() - 1; }
        size_t const P = pieceBase[S];
        ZsaPieceFlags arrived(P);
        std::vector<std::thread> helpers;
        auto const work = [&](unsigned w) {
            for (size_t p = w; p < P; p += T) {
                if (arrived.wait(p) < 0) continue;             // (the transfer failed: nothing to hand out)
                unsigned k = 0; while (p >= pieceBase[k + 1]) k++;
                size_t const q = p - pieceBase[k];
                const u8* const from = hFrom[k];
                for (size_t i = cuts[k][q]; i < cuts[k][q + 1]; i++) if (!zsa_isError(outSize[i]) && outSize[i]) memcpy(dst[i], from + hostOff[i], outSize[i]);
            }
        };
        unsigned made = 0;
        try { for (; made < T; made++) helpers.emplace_back(work, made); } catch (...) { }
        for (unsigned k = 0; k < S; k++) {
            hipStream_t const q = st.stream[k];
            size_t const a = sub[k], b = sub[k + 1];
            size_t e = firstError ? firstError : zsa_batch_results(st.batch[k], outSize + a, q);
            ZSA_TRACE("sub-batch %u results", k);
            hipError_t err = hipSuccess;
            if (!e) {
                size_t produced = 0, span = 0;
                for (size_t i = a; i < b; i++) { finish(i); if (!zsa_isError(outSize[i]) && outSize[i]) { produced += (outSize[i] + 15) & ~(size_t)15; span = outOff[i] + outSize[i] - outOff[a]; } }
                // where the frames filled less than three quarters of their slots (compressed output), the produced bytes are first
                // packed into this sub-batch's part of the input staging buffer -- its content is spent -- and travel from there
                size_t const inLo = inOff[a], inHi = inOff[b];
                // (the host side of the packed bytes is the input staging buffer, or the output one when the payloads came from pinned memory)
                bool const pack = produced <= inHi - inLo && produced <= outOff[b] - outOff[a] && produced < span / 4 * 3;
                const u8* d_from; u8* h_to;
                if (pack) {
                    size_t at = inLo;
                    for (size_t i = a; i < b; i++) {
                        size_t const sz = (!zsa_isError(outSize[i])) ? outSize[i] : 0;
                        hostOff[i] = at; h_pk[3 * i] = outOff[i]; h_pk[3 * i + 1] = at; h_pk[3 * i + 2] = sz;
                        at += (sz + 15) & ~(size_t)15;
                    }
                    err = hipMemcpyAsync(d_pk + 3 * a, h_pk + 3 * a, 3 * (b - a) * sizeof(u64), hipMemcpyHostToDevice, q);
                    if (err == hipSuccess) { hipLaunchKernelGGL(zsa_k_pack, dim3((unsigned)(b - a)), dim3(256), 0, q, (const u8*)st.d_out, st.d_in, (const u64*)(d_pk + 3 * a), (u32)(b - a)); err = hipGetLastError(); }
                    d_from = st.d_in;
                    h_to = pinIn ? st.h_out + outOff[a] - inLo : st.h_in;      // (device offset x <-> host offset x of h_to)
                } else {
                    for (size_t i = a; i < b; i++) hostOff[i] = outOff[i];
                    d_from = st.d_out; h_to = st.h_out;
                }
                hFrom[k] = h_to;
                for (size_t p = pieceBase[k]; p < pieceBase[k + 1]; p++) {
                    size_t const c0 = cuts[k][p - pieceBase[k]], c1 = cuts[k][p - pieceBase[k] + 1];
                    // the piece's frames lie in [lo, hi) of the buffer they travel through
                    size_t lo = (size_t)-1, hi = 0;
                    for (size_t i = c0; i < c1; i++) if (!zsa_isError(outSize[i]) && outSize[i]) { if (hostOff[i] < lo) lo = hostOff[i]; if (hostOff[i] + outSize[i] > hi) hi = hostOff[i] + outSize[i]; }
                    if (err == hipSuccess && hi > lo) { err = hipMemcpyAsync(h_to + lo, d_from + lo, hi - lo, hipMemcpyDeviceToHost, q); if (err == hipSuccess) err = hipStreamSynchronize(q); }
                    arrived.set(p, err == hipSuccess ? 1 : -1);
                }
                if (err != hipSuccess) { fprintf(stderr, "libzstd_amd: host batch download failed: %s\n", hipGetErrorString(err)); (void)hipGetLastError(); e = zsa_fail(ZSA_EF_RUNTIME, kErrGeneric); }
            } else {
                for (size_t p = pieceBase[k]; p < pieceBase[k + 1]; p++) arrived.set(p, -1);
            }
            if (e && !firstError) firstError = e;
            ZSA_TRACE("sub-batch %u downloaded", k);
        }
        for (unsigned w = made; w < T; w++) work(w);   // (a helper that could not be started: its share is handed out here)
        for (auto& h : helpers) h.join();
        ZSA_TRACE("handed out");
    }
    return firstError;
}

// ---- one host batch over several GPUs.  The reference's callers use a whole node by fanning payloads out over goroutines
// (zstd_stream_test.go:149-169); a process that hands ONE batch to this library gets the same from ZSTD_AMD_DEVICES=<n|all>:
// a large batch without a dictionary (dictionary digests live on one device) is cut into contiguous shares of about equal
// bytes, one per device, each run by a worker thread that stays bound to its device and keeps its staging from call to call.
// Default 1: the calling thread's current device only.
struct ZsaDevWorker {
    std::thread th; std::mutex m; std::condition_variable cv;
    int device = 0; bool busy = false, quit = false;
    std::function<void()> job;
};
static std::mutex g_devMutex;                      // one multi-device batch at a time (the workers are shared)
static std::vector<ZsaDevWorker*> g_devWorkers;
static unsigned zsa_batch_devices()
{
    static unsigned const conf = [] {
        const char* e = getenv("ZSTD_AMD_DEVICES");
        if (!e) return 1u;
        int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
        if (strcmp(e, "all") == 0) return (unsigned)(n > 0 ? n : 1);
        int const v = atoi(e);
        // (ZSTD_AMD_DEVICE_ALIAS=1, a test knob: the shares all run on the current device, so that the split can be exercised on a one-GPU box)
        if (getenv("ZSTD_AMD_DEVICE_ALIAS")) return (unsigned)(v < 1 ? 1 : (v > 8 ? 8 : v));
        return (unsigned)(v < 1 ? 1 : (v > n ? (n > 0 ? n : 1) : v));
    }();
    return conf;
}
static void zsa_dev_worker_main(ZsaDevWorker* w)
{
    if (hipSetDevice(w->device) != hipSuccess) (void)hipGetLastError();
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->busy || w->quit; });
        if (w->quit) break;
        lk.unlock(); w->job(); lk.lock();
        w->busy = false;
        w->cv.notify_all();
    }
}
static size_t zsa_run_host_batch_devices(bool compress, const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity,
                                         size_t* outSize, size_t n, int level, const zsa_CDict* cd, const zsa_DDict* dd)
{
    unsigned D = (cd || dd || n < 64) ? 1u : zsa_batch_devices();
    size_t total = 0;
    if (D > 1) { for (size_t i = 0; i < n; i++) total += srcSize[i] + dstCapacity[i]; if (total < ((size_t)64 << 20)) D = 1; }
    if (D <= 1) return zsa_run_host_batch(compress, src, srcSize, dst, dstCapacity, outSize, n, level, cd, dd);
    int cur = 0, count = 0;
    if (hipGetDevice(&cur) != hipSuccess || hipGetDeviceCount(&count) != hipSuccess || count < 1) { (void)hipGetLastError(); return zsa_fail(ZSA_EF_NO_DEVICE, kErrGeneric); }
    bool const alias = getenv("ZSTD_AMD_DEVICE_ALIAS") != nullptr;
    std::lock_guard<std::mutex> g(g_devMutex);
    try {
        while (g_devWorkers.size() < D) {
            ZsaDevWorker* w = new ZsaDevWorker();
            w->device = alias ? cur : (int)((cur + g_devWorkers.size()) % (size_t)count);
            w->th = std::thread(zsa_dev_worker_main, w);
            w->th.detach();                        // (lives as long as the process: bound to its device, holding its staging)
            g_devWorkers.push_back(w);
        }
    } catch (...) { return zsa_run_host_batch(compress, src, srcSize, dst, dstCapacity, outSize, n, level, cd, dd); }
    // shares of about equal bytes: share d = frames [cut[d], cut[d + 1])
    std::vector<size_t> cut(D + 1, n);
    {   size_t acc = 0, d = 1; cut[0] = 0;
        for (size_t i = 0; i < n && d < D; i++) { acc += srcSize[i] + dstCapacity[i]; if (acc >= total / D * d) cut[d++] = i + 1; }
    }
    std::vector<size_t> rc(D, 0); std::vector<int> why(D, 0);
    for (unsigned d = 0; d < D; d++) {
        ZsaDevWorker* w = g_devWorkers[d];
        size_t const a = cut[d], b = cut[d + 1];
        std::lock_guard<std::mutex> lk(w->m);
        w->job = [=, &rc, &why] {
            rc[d] = b > a ? zsa_run_host_batch(compress, src + a, srcSize + a, dst + a, dstCapacity + a, outSize + a, b - a, level, nullptr, nullptr) : 0;
            why[d] = t_engineFailure;
        };
        w->busy = true;
        w->cv.notify_all();
    }
    size_t first = 0;
    for (unsigned d = 0; d < D; d++) {
        ZsaDevWorker* w = g_devWorkers[d];
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return !w->busy; });
        if (rc[d] && !first) { first = rc[d]; t_engineFailure = why[d]; }
    }
    return first;
}

extern "C" size_t zsa_compress_batch(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity,
                                     size_t* outSize, size_t n, int level, const zsa_CDict* cdict)
{ t_engineFailure = ZSA_EF_NONE; return zsa_run_host_batch_devices(true, src, srcSize, dst, dstCapacity, outSize, n, level, cdict, nullptr); }
extern "C" size_t zsa_decompress_batch(const void* const* src, const size_t* srcSize, void* const* dst, const size_t* dstCapacity,
                                       size_t* outSize, size_t n, const zsa_DDict* ddict)
{ t_engineFailure = ZSA_EF_NONE; return zsa_run_host_batch_devices(false, src, srcSize, dst, dstCapacity, outSize, n, 0, nullptr, ddict); }

extern "C" size_t zsa_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level)
{
    size_t out = 0;
    size_t const e = zsa_compress_batch(&src, &srcSize, &dst, &dstCapacity, &out, 1, level, nullptr);
    return e ? e : out;
}
extern "C" size_t zsa_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    size_t out = 0;
    size_t const e = zsa_decompress_batch(&src, &srcSize, &dst, &dstCapacity, &out, 1, nullptr);
    return e ? e : out;
}
