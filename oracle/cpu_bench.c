/* cpu_bench.c -- TEST/BENCH INFRASTRUCTURE: times a CPU Zstandard implementation on the GPU box's host cores,
 * one call per frame from N pthreads over disjoint shards with preallocated dst (the C
 * stand-in for the Go goroutine driver, since `go` is not installed).
 *   cpu_bench <lib.so> <ref|oracle> <kind 0|1> <frameSize> <nFrames> <seed> <threads> <level>
 * lib = oracle/_ref/libzstd_ref.so (the reference's own C, kind "reference") or oracle/liboracle.so ("port").
 * Prints one JSON object.  Never linked into the product.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

void zc_generate(uint8_t* dst, size_t frameSize, uint64_t firstFrame, uint64_t nFrames, uint64_t seed, int kind, int nThreads);

typedef size_t (*compress_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*decompress_fn)(void*, size_t, const void*, size_t);
typedef size_t (*bound_fn)(size_t);

static compress_fn g_c; static decompress_fn g_d; static bound_fn g_b;
static uint8_t *g_src, *g_cbuf, *g_out;
static size_t g_frame, g_n, g_bound; static size_t* g_csize; static int g_level, g_threads, g_fail;
static int g_phase;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void* run(void* a)
{
    size_t t = (size_t)a;
    size_t lo = g_n * t / (size_t)g_threads, hi = g_n * (t + 1) / (size_t)g_threads;
    for (size_t i = lo; i < hi; i++) {
        if (g_phase == 0) {
            size_t r = g_c(g_cbuf + i * g_bound, g_bound, g_src + i * g_frame, g_frame, g_level);
            if (r > g_bound) g_fail = 1;
            g_csize[i] = r;
        } else {
            size_t r = g_d(g_out + i * g_frame, g_frame, g_cbuf + i * g_bound, g_csize[i]);
            if (r != g_frame) g_fail = 1;
        }
    }
    return NULL;
}
static double phase(int p)
{
    pthread_t th[256];
    g_phase = p;
    double t0 = now();
    for (int t = 0; t < g_threads; t++) pthread_create(&th[t], NULL, run, (void*)(size_t)t);
    for (int t = 0; t < g_threads; t++) pthread_join(th[t], NULL);
    return now() - t0;
}
int main(int argc, char** argv)
{
    if (argc < 9) { fprintf(stderr, "usage: cpu_bench lib ref|oracle kind frameSize nFrames seed threads level\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    int isRef = strcmp(argv[2], "ref") == 0;
    g_c = (compress_fn)dlsym(h, isRef ? "ZSTD_compress" : "zo_compress");
    g_d = (decompress_fn)dlsym(h, isRef ? "ZSTD_decompress" : "zo_decompress");
    g_b = (bound_fn)dlsym(h, isRef ? "ZSTD_compressBound" : "zo_compressBound");
    const char* (*ver)(void) = (const char* (*)(void))dlsym(h, "ZSTD_versionString");
    if (!g_c || !g_d || !g_b) { fprintf(stderr, "missing symbols\n"); return 2; }
    int kind = atoi(argv[3]); g_frame = (size_t)atoll(argv[4]); g_n = (size_t)atoll(argv[5]);
    uint64_t seed = (uint64_t)atoll(argv[6]); g_threads = atoi(argv[7]); g_level = atoi(argv[8]);
    if (g_threads < 1) g_threads = 1; if (g_threads > 256) g_threads = 256;
    g_bound = g_b(g_frame);
    g_src = malloc(g_frame * g_n); g_cbuf = malloc(g_bound * g_n); g_out = malloc(g_frame * g_n); g_csize = malloc(sizeof(size_t) * g_n);
    if (!g_src || !g_cbuf || !g_out || !g_csize) { fprintf(stderr, "oom\n"); return 2; }
    zc_generate(g_src, g_frame, 0, g_n, seed, kind, g_threads > 64 ? 64 : g_threads);
    memset(g_cbuf, 0, g_bound * g_n); memset(g_out, 0, g_frame * g_n);
    double tc = phase(0);
    double td = phase(1);
    size_t csum = 0; for (size_t i = 0; i < g_n; i++) csum += g_csize[i];
    int same = memcmp(g_src, g_out, g_frame * g_n) == 0;
    double bytes = (double)g_frame * (double)g_n;
    printf("{\"impl\":\"%s\",\"libzstd\":\"%s\",\"threads\":%d,\"frames\":%zu,\"frame_size\":%zu,\"compress_GBps\":%.4f,\"decompress_GBps\":%.4f,\"ratio\":%.4f,\"roundtrip_ok\":%s}\n",
           argv[2], ver ? ver() : "n/a", g_threads, g_n, g_frame, bytes / tc / 1e9, bytes / td / 1e9, bytes / (double)csum, (same && !g_fail) ? "true" : "false");
    return (same && !g_fail) ? 0 : 1;
}
