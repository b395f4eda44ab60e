/* abi_smoke.c -- the C ABI of libzstd_amd.so called from plain C, the way the cgo stubs in shim/ call it.
 *
 * This image has no Go toolchain, so the Go files under shim/ cannot be built here; this program makes the same calls
 * in the same order with the same argument conventions (host pointers, sizes, libzstd's size_t-or-negative-error
 * returns) and is what stands in for `go test -tags zstd_amd`.  Built by __graft_entry__.build() into tests/abi_smoke;
 * run by tests/test_abi_smoke.py:
 *     abi_smoke            header-only calls (no GPU needed): bounds, error strings, frame-header peek, parameters
 *     abi_smoke gpu        plus the calls that move data: one-shot, dictionary, host batch (pageable and pinned buffers)
 *     abi_smoke enginefail the fallback decision of shim/zstd_amd_on.go: an engine-side failure (no usable device; or, with
 *                          ZSTD_AMD_FAIL=alloc in the environment, every workspace allocation refused) comes back as an error
 *                          code AND is flagged by zsa_engine_failure(), and the binding answers with its CPU path; a data error
 *                          is not flagged.  Runs with or without a GPU.
 * Exit status 0 = every check held; a failed check prints its line and exits 1.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zstd_amd.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "abi_smoke: line %d: check failed: %s\n", __LINE__, #c); exit(1); } } while (0)

static void fill(unsigned char* p, size_t n, unsigned seed)
{
    static const char* words[] = { "GET ", "POST ", "/api/v1/series ", "status=200 ", "status=404 ", "host=web-", "latency_ms=", "\n" };
    size_t o = 0;
    while (o < n) {
        seed = seed * 1103515245u + 12345u;
        const char* w = words[(seed >> 16) & 7];
        size_t l = strlen(w);
        if (l > n - o) l = n - o;
        memcpy(p + o, w, l); o += l;
        if (o < n && ((seed >> 20) & 3) == 0) p[o++] = (unsigned char)('0' + ((seed >> 24) % 10));
    }
}

static void header_only(void)
{
    CHECK(zsa_version() != NULL && strlen(zsa_version()) > 0);
    /* zstd.go:57 CompressBound: ZSTD_COMPRESSBOUND arithmetic */
    CHECK(zsa_compressBound(0) == 64);
    CHECK(zsa_compressBound(131072) == 131072 + (131072 >> 8));
    CHECK(zsa_compressBound(1000) == 1000 + (1000 >> 8) + (((128u << 10) - 1000) >> 11));
    /* errors.go:13-34: codes and the strings IsDstSizeTooSmallError matches */
    CHECK(!zsa_isError(0) && !zsa_isError(12345));
    CHECK(zsa_isError((size_t)-70) && strcmp(zsa_getErrorName((size_t)-70), "Destination buffer is too small") == 0);
    CHECK(strcmp(zsa_getErrorName((size_t)-40), "Unsupported parameter") == 0);
    CHECK(strcmp(zsa_getErrorName((size_t)-10), "Unknown frame descriptor") == 0);
    CHECK(strcmp(zsa_getErrorName((size_t)-20), "Data corruption detected") == 0);
    CHECK(strcmp(zsa_getErrorName(0), "No error detected") == 0);
    /* zstd.go:71 decompressSizeHint: frame-header peek.  Magic, single-segment descriptor with a 1-byte content size. */
    { unsigned char h[6] = { 0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x0C };
      CHECK(zsa_getFrameContentSize(h, sizeof h) == 12);
      CHECK(zsa_getFrameContentSize(h, 3) == (unsigned long long)-2);            /* ZSTD_CONTENTSIZE_ERROR */
      h[4] = 0x00; h[5] = 0x58;                                                  /* window descriptor, no content size */
      CHECK(zsa_getFrameContentSize(h, sizeof h) == (unsigned long long)-1); }   /* ZSTD_CONTENTSIZE_UNKNOWN */
    /* parameters of level 1 at 128 KiB (clevels.h, <= 128 KB table, row 1): ZSTD_fast, windowLog 17 */
    { unsigned cp[7];
      CHECK(zsa_getCParams(1, 131072, 0, 0, cp) == 0);
      CHECK(cp[0] == 17 && cp[6] == 1);
      CHECK(zsa_getCParams(19, 131072, 0, 0, cp) == 1); }                        /* btultra2: outside the engine */
}

static void moving_data(void)
{
    enum { N = 48, MAXLEN = 200000 };
    CHECK(zsa_device_count() > 0);
    unsigned char* src = malloc(MAXLEN);
    size_t const bound = zsa_compressBound(MAXLEN);
    unsigned char* c1 = malloc(bound), * back = malloc(MAXLEN + 1);
    fill(src, MAXLEN, 7);

    /* zstd.go:108,115 / :168 -- one-shot, including the empty payload (NULL source pointer, zstd.go:103-105) */
    static const size_t lens[] = { 0, 1, 11, 1000, 131072, MAXLEN };
    for (unsigned k = 0; k < sizeof lens / sizeof lens[0]; k++) {
        size_t n = lens[k];
        size_t c = zsa_compress(c1, zsa_compressBound(n), n ? src : NULL, n, 1);
        CHECK(!zsa_isError(c));
        CHECK(zsa_getFrameContentSize(c1, c) == n);
        size_t d = zsa_decompress(back, n + 1, c1, c);
        CHECK(d == n && memcmp(back, src, n) == 0);
        if (n > 1) {   /* destination too small: the reference's error, by name (errors.go:30-34) */
            size_t e = zsa_decompress(back, n - 1, c1, c);
            CHECK(zsa_isError(e) && strcmp(zsa_getErrorName(e), "Destination buffer is too small") == 0);
        }
    }
    /* a level outside the engine: code 40, which the Go hook answers with the vendored C (shim/zstd_amd_on.go) */
    CHECK(zsa_compress(c1, bound, src, 1000, 19) == (size_t)-40);
    /* garbage: "Unknown frame descriptor" */
    CHECK(zsa_decompress(back, MAXLEN, src, 100) == (size_t)-10);

    /* zstd_bulk.go:41-52,77,86,125,144-151 -- dictionary life cycle */
    { unsigned char dict[4096]; fill(dict, sizeof dict, 99);
      zsa_CDict* cd = zsa_createCDict(dict, sizeof dict, 1);
      zsa_DDict* dd = zsa_createDDict(dict, sizeof dict);
      CHECK(cd != NULL && dd != NULL);
      size_t c = zsa_compress_usingCDict(c1, zsa_compressBound(3000), src, 3000, cd);
      CHECK(!zsa_isError(c));
      size_t plain = zsa_compress(back, zsa_compressBound(3000), src, 3000, 1);
      CHECK(c < plain);                                        /* the dictionary helped */
      CHECK(zsa_decompress_usingDDict(back, 3000, c1, c, dd) == 3000 && memcmp(back, src, 3000) == 0);
      CHECK(zsa_isError(zsa_decompress(back, 3000, c1, c)) || memcmp(back, src, 3000) != 0);   /* and is needed */
      CHECK(zsa_freeCDict(cd) == 0 && zsa_freeDDict(dd) == 0); }

    /* new surface: host batch (shim/zstd_amd_batch.go) == the single calls, frame by frame */
    { const void* srcs[N]; size_t srcSizes[N]; void* dsts[N]; size_t caps[N]; size_t out[N];
      for (int i = 0; i < N; i++) {
          srcSizes[i] = (size_t)((i * 7919u) % 150000u); if (i == 0) srcSizes[i] = 0;
          srcs[i] = srcSizes[i] ? src + (i * 131u) % 40000u : NULL;
          caps[i] = zsa_compressBound(srcSizes[i]); dsts[i] = malloc(caps[i]); }
      CHECK(zsa_compress_batch(srcs, srcSizes, dsts, caps, out, N, 1, NULL) == 0);
      for (int i = 0; i < N; i++) {
          size_t c = zsa_compress(c1, caps[i], srcs[i], srcSizes[i], 1);
          CHECK(out[i] == c && memcmp(dsts[i], c1, c) == 0); }
      const void* fsrc[N]; size_t fsz[N]; void* bdst[N]; size_t bcap[N]; size_t bout[N];
      for (int i = 0; i < N; i++) { fsrc[i] = dsts[i]; fsz[i] = out[i]; bcap[i] = srcSizes[i] + 1; bdst[i] = malloc(bcap[i]); }
      CHECK(zsa_decompress_batch(fsrc, fsz, bdst, bcap, bout, N, NULL) == 0);
      for (int i = 0; i < N; i++) {
          CHECK(bout[i] == srcSizes[i] && (srcSizes[i] == 0 || memcmp(bdst[i], srcs[i], srcSizes[i]) == 0));
          free(dsts[i]); free(bdst[i]); } }
    zsa_thread_release();
    free(src); free(c1); free(back);
}

/* hotCompress / hotDecompress of shim/zstd_amd_on.go, with the vendored call replaced by a marker: returns 1 when the binding
 * would answer this call with the vendored C, 0 when it hands the engine's answer to the caller */
static int hook_falls_back(size_t r, int legacy_magic)
{
    if (r == (size_t)-40) return 1;                           /* errParameterUnsupported */
    if (r == (size_t)-10 && legacy_magic) return 1;           /* errPrefixUnknown on a legacy frame */
    if (zsa_isError(r) && zsa_engine_failure() != 0) return 1;
    return 0;
}

static void engine_failure(void)
{
    unsigned char src[3000], dst[4096], back[3000];
    fill(src, sizeof src, 5);
    int const have_gpu = zsa_device_count() > 0;
    const char* forced = getenv("ZSTD_AMD_FAIL");
    if (!have_gpu || (forced && strcmp(forced, "alloc") == 0)) {
        /* the engine cannot run: every data call fails, is flagged, and the hook falls back */
        size_t c = zsa_compress(dst, sizeof dst, src, sizeof src, 1);
        CHECK(zsa_isError(c) && zsa_engine_failure() != 0 && hook_falls_back(c, 0));
        printf("compress: engine code \"%s\", zsa_engine_failure() = %d -> vendored C\n", zsa_getErrorName(c), zsa_engine_failure());
        unsigned char frame[13] = { 0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x04, 0x21, 0x00, 0x00, 'a', 'b', 'c', 'd' };   /* raw block "abcd" */
        size_t d = zsa_decompress(back, sizeof back, frame, sizeof frame);
        CHECK(zsa_isError(d) && zsa_engine_failure() != 0 && hook_falls_back(d, 0));
        printf("decompress: engine code \"%s\", zsa_engine_failure() = %d -> vendored C\n", zsa_getErrorName(d), zsa_engine_failure());
        /* batch: the whole call fails and is flagged; the binding answers payload by payload (shim/zstd_amd_batch.go) */
        const void* srcs[2] = { src, src + 100 }; size_t sz[2] = { 1000, 1000 }; void* dsts[2] = { dst, dst + 2048 }; size_t caps[2] = { 2048, 2048 }; size_t out[2];
        size_t rc = zsa_compress_batch(srcs, sz, dsts, caps, out, 2, 1, NULL);
        CHECK(zsa_isError(rc) && zsa_engine_failure() != 0);
        printf("batch: engine code \"%s\", zsa_engine_failure() = %d -> vendored C per payload\n", zsa_getErrorName(rc), zsa_engine_failure());
        if (!have_gpu) {   /* dictionaries: no device digest, the processor keeps the vendored dictionaries (newBulkAccel) */
            CHECK(zsa_createDDict(src, 64) == NULL && zsa_engine_failure() != 0);
            CHECK(zsa_batch_create(4) == NULL && zsa_engine_failure() != 0);
        }
    } else {
        /* a working engine: data errors are the caller's to see, and are not flagged */
        size_t c = zsa_compress(dst, sizeof dst, src, sizeof src, 1);
        CHECK(!zsa_isError(c) && zsa_engine_failure() == 0);
        size_t d = zsa_decompress(back, 10, dst, c);                                   /* destination too small */
        CHECK(d == (size_t)-70 && zsa_engine_failure() == 0 && !hook_falls_back(d, 0));
        d = zsa_decompress(back, sizeof back, src, 100);                              /* not a frame */
        CHECK(d == (size_t)-10 && zsa_engine_failure() == 0 && !hook_falls_back(d, 0));
        c = zsa_compress(dst, sizeof dst, src, sizeof src, 19);                       /* a level outside the engine */
        CHECK(c == (size_t)-40 && zsa_engine_failure() == 0 && hook_falls_back(c, 0));
        printf("data errors are not flagged; code 40 falls back\n");
    }
}

/* ZSTD_compress serves any capacity the frame fits (the host entry points do the same); pinned buffers from zsa_host_alloc
 * are read and written by the GPU in place */
static void capacity_and_pinned(void)
{
    enum { N = 96, LEN = 131072 };
    unsigned char* src = malloc(LEN);
    fill(src, LEN, 11);
    size_t const bound = zsa_compressBound(LEN);
    unsigned char* ref = malloc(bound);
    size_t const c = zsa_compress(ref, bound, src, LEN, 1);
    CHECK(!zsa_isError(c) && c < bound / 2);
    { unsigned char* small = malloc(c);
      CHECK(zsa_compress(small, c, src, LEN, 1) == c && memcmp(small, ref, c) == 0);            /* exactly fits */
      CHECK(zsa_compress(small, c - 1, src, LEN, 1) == (size_t)-70);                            /* one byte short */
      free(small); }
    /* a 12 MiB batch (large path: helper threads, sub-batches) from pageable and from pinned memory gives the same frames */
    unsigned char* slab = zsa_host_alloc((size_t)N * LEN);
    unsigned char* cslab = zsa_host_alloc((size_t)N * bound);
    unsigned char* bslab = zsa_host_alloc((size_t)N * LEN);
    CHECK(slab && cslab && bslab);
    unsigned char* pg = malloc((size_t)N * LEN); unsigned char* cpg = malloc((size_t)N * bound);
    const void* srcs[N]; const void* psrcs[N]; size_t sz[N]; void* dsts[N]; void* pdsts[N]; size_t caps[N]; size_t out[N], pout[N];
    for (int i = 0; i < N; i++) {
        fill(pg + (size_t)i * LEN, LEN, 100 + i);
        sz[i] = LEN - (size_t)(i * 37 % 5000); if (i == 3) sz[i] = 0;
        memcpy(slab + (size_t)i * LEN + (i & 1), pg + (size_t)i * LEN, sz[i] ? sz[i] - 1 : 0);      /* odd addresses on the pinned side */
        memcpy(pg + (size_t)i * LEN, slab + (size_t)i * LEN + (i & 1), sz[i] ? sz[i] - 1 : 0);
        if (sz[i]) sz[i] -= 1;
        srcs[i] = pg + (size_t)i * LEN; psrcs[i] = slab + (size_t)i * LEN + (i & 1);
        dsts[i] = cpg + (size_t)i * bound; pdsts[i] = cslab + (size_t)i * bound + (i % 3); caps[i] = bound - 3;
    }
    CHECK(zsa_compress_batch(srcs, sz, dsts, caps, out, N, 1, NULL) == 0);
    CHECK(zsa_compress_batch(psrcs, sz, pdsts, caps, pout, N, 1, NULL) == 0);
    for (int i = 0; i < N; i++) CHECK(!zsa_isError(out[i]) && out[i] == pout[i] && memcmp(dsts[i], pdsts[i], out[i]) == 0);
    /* pinned in, pageable out and the reverse */
    CHECK(zsa_compress_batch(psrcs, sz, dsts, caps, pout, N, 1, NULL) == 0);
    for (int i = 0; i < N; i++) CHECK(out[i] == pout[i] && memcmp(dsts[i], pdsts[i], out[i]) == 0);
    /* decompress: pinned frames into pinned destinations, and into pageable ones */
    void* bd[N]; void* pbd[N]; size_t bcap[N], bout[N], pbout[N];
    unsigned char* bpg = malloc((size_t)N * LEN);
    for (int i = 0; i < N; i++) { bd[i] = bpg + (size_t)i * LEN; pbd[i] = bslab + (size_t)i * LEN; bcap[i] = sz[i]; }
    CHECK(zsa_decompress_batch((const void* const*)pdsts, out, pbd, bcap, pbout, N, NULL) == 0);
    CHECK(zsa_decompress_batch((const void* const*)dsts, out, bd, bcap, bout, N, NULL) == 0);
    for (int i = 0; i < N; i++) {
        CHECK(bout[i] == sz[i] && pbout[i] == sz[i]);
        CHECK(sz[i] == 0 || (memcmp(bd[i], srcs[i], sz[i]) == 0 && memcmp(pbd[i], srcs[i], sz[i]) == 0));
    }
    /* a destination that is too small, in the pinned path: the frame's error, neighbours untouched */
    bcap[5] = sz[5] - 1; memset(pbd[5], 0xEE, sz[5]);
    CHECK(zsa_decompress_batch((const void* const*)pdsts, out, pbd, bcap, pbout, N, NULL) == 0);
    CHECK(pbout[5] == (size_t)-70 && pbout[4] == sz[4] && pbout[6] == sz[6] && memcmp(pbd[6], srcs[6], sz[6]) == 0);
    CHECK(zsa_noalloc_prepare() == 0);
    zsa_host_free(slab); zsa_host_free(cslab); zsa_host_free(bslab);
    free(pg); free(cpg); free(bpg); free(src); free(ref);
    zsa_thread_release();
}

int main(int argc, char** argv)
{
    header_only();
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) { moving_data(); capacity_and_pinned(); }
    if (argc > 1 && strcmp(argv[1], "enginefail") == 0) engine_failure();
    printf("abi_smoke ok (%s)\n", argc > 1 ? argv[1] : "header-only");
    return 0;
}
