/* zstd_oracle.h -- public entry points of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product library; used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.  Parity is pinned against the
 * reference's own C (oracle/_ref/libzstd_ref.so, built by oracle/Makefile from /root/reference) and the
 * fixtures in tests/golden/.
 *
 * Return convention == libzstd's (zstd.h, error_private.h:57-61): bytes written, or (size_t)-errorCode.
 */
#ifndef ZSTD_ORACLE_H
#define ZSTD_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

size_t zo_compressBound(size_t srcSize);                                   /* zstd.h ZSTD_COMPRESSBOUND */
size_t zo_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level);   /* ZSTD_compress */
size_t zo_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);            /* ZSTD_decompress */
unsigned long long zo_getFrameContentSize(const void* src, size_t srcSize);                       /* ZSTD_getFrameContentSize */
unsigned zo_isError(size_t code);                                          /* ZSTD_isError */
const char* zo_getErrorName(size_t code);                                  /* ZSTD_getErrorName */

/* dictionary path (zstd_bulk.go:33-141) */
typedef struct zo_cdict_s zo_cdict_t;
typedef struct zo_ddict_s zo_ddict_t;
zo_cdict_t* zo_createCDict(const void* dict, size_t dictSize, int level);  /* ZSTD_createCDict */
void        zo_freeCDict(zo_cdict_t*);
size_t zo_compress_usingCDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zo_cdict_t*);
zo_ddict_t* zo_createDDict(const void* dict, size_t dictSize);             /* ZSTD_createDDict */
void        zo_freeDDict(zo_ddict_t*);
size_t zo_decompress_usingDDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zo_ddict_t*);

/* introspection used by tests to localise a divergence */
typedef struct { uint32_t offBase; uint16_t litLength; uint16_t mlBase; } zo_seq_t;
typedef void (*zo_seq_dump_fn)(void* arg, const void* seqs /* zo_seq_t[] */, uint32_t nbSeq,
                               const uint8_t* literals, size_t litSize, const uint32_t rep[3]);
size_t zo_compress_dump(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level,
                        zo_seq_dump_fn dump, void* dumpArg);
int zo_cparams_query(int level, unsigned long long srcSize, size_t dictSize, int mode, unsigned out[7]);

#ifdef __cplusplus
}
#endif
#endif
