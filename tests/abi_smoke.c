/* abi_smoke.c -- the C ABI of libzstd_amd.so called from plain C, the way the cgo stubs in shim/ call it.
 *
 * This image has no Go toolchain, so the Go files under shim/ cannot be built here; this program makes the same calls
 * in the same order with the same argument conventions (host pointers, sizes, libzstd's size_t-or-negative-error
 * returns) and is what stands in for `go test -tags zstd_amd`.  Built by __graft_entry__.build() into tests/abi_smoke;
 * run by tests/test_abi_smoke.py:
 *     abi_smoke            header-only calls (no GPU needed): bounds, error strings, frame-header peek, parameters
 *     abi_smoke gpu        plus the calls that move data: one-shot, dictionary, host batch
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

int main(int argc, char** argv)
{
    header_only();
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) moving_data();
    printf("abi_smoke ok (%s)\n", argc > 1 ? argv[1] : "header-only");
    return 0;
}
