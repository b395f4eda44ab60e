// emu_driver.cpp -- TEST-ONLY: runs the HIP kernel bodies on the CPU, one emulated 64-lane wave at a time.
// Built by tests/emu/Makefile into tests/emu/libzsa_emu.so and driven from tests/test_emu_*.py.
// Never part of the product; the product library has no CPU path.
#define ZSA_WAVE_HEADER "zsa_wave_emu.h"
extern "C" { unsigned long long g_zsa_stat[8]; }
#define ZSA_STAT(i) do { if (lane == 0) g_zsa_stat[i]++; } while (0)
extern "C" { void (*g_zsa_seq_hook)(const unsigned long long* seq, unsigned n) = nullptr; }
#define ZSA_SEQ_HOOK(seq, n) do { wv_sync(); if (lane == 0 && g_zsa_seq_hook) g_zsa_seq_hook((const unsigned long long*)(seq), (n)); wv_sync(); } while (0)
#include "zsa_common.h"
#include "zsa_decode.h"
#ifdef ZSA_EMU_ENCODE
#include "zsa_encode.h"
#include "zsa_dict.h"
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

ZsaEmu g_emu;
void emu_divergence(int lineA, int lineB, int lane)
{
    fprintf(stderr, "emulator: divergent wave primitive: some lanes at source line %d, lane %d at line %d\n", lineA, lane, lineB);
    abort();
}
void emu_not_uniform(int line)
{
    fprintf(stderr, "emulator: wv_uniform() at source line %d was given a value that differs between lanes\n", line);
    abort();
}
static emu_kernel_fn g_fn; static void* g_arg;
static void lane_entry()
{
    g_fn(g_arg);
    int me = g_emu.cur;
    g_emu.done[me] = 1; g_emu.ndone++;
    if (g_emu.arrived && g_emu.arrived == 64 - g_emu.ndone) { g_emu.arrived = 0; g_emu.gen++; }
    // hand over to the next unfinished lane, or back to main
    for (int k = 1; k <= 64; k++) { int n = (me + k) & 63; if (!g_emu.done[n]) { g_emu.cur = n; swapcontext(&g_emu.lane_ctx[me], &g_emu.lane_ctx[n]); } }
    setcontext(&g_emu.main_ctx);
}
void emu_run_wave(emu_kernel_fn fn, void* arg)
{
    const size_t STK = 256 * 1024;
    if (!g_emu.stacks) g_emu.stacks = (char*)malloc(64 * STK);
    g_fn = fn; g_arg = arg;
    g_emu.arrived = 0; g_emu.gen = 0; g_emu.ndone = 0;
    for (int i = 0; i < 64; i++) {
        g_emu.done[i] = 0;
        getcontext(&g_emu.lane_ctx[i]);
        g_emu.lane_ctx[i].uc_stack.ss_sp = g_emu.stacks + i * STK;
        g_emu.lane_ctx[i].uc_stack.ss_size = STK;
        g_emu.lane_ctx[i].uc_link = &g_emu.main_ctx;
        makecontext(&g_emu.lane_ctx[i], lane_entry, 0);
    }
    g_emu.cur = 0;
    swapcontext(&g_emu.main_ctx, &g_emu.lane_ctx[0]);
}

struct DecArgs { ZsaDecShared* S; const u8* src; u64 srcSize; u8* dst; u64 cap; u8* litBuf; const ZsaDDictDev* dd; u64 result; };
static void dec_entry(void* a)
{
    DecArgs* A = (DecArgs*)a;
    u64 r = zsa_decode_frames(A->S, A->src, A->srcSize, A->dst, A->cap, A->litBuf, A->dd);
    if (wv_lane() == 0) A->result = r;
}
extern "C" size_t emu_decompress(void* dst, size_t cap, const void* src, size_t n)
{
    static ZsaDecShared S;
    static u8* litBuf = (u8*)malloc(ZSA_DEC_WKSP_BYTES + 64);
    DecArgs A = { &S, (const u8*)src, n, (u8*)dst, cap, litBuf, nullptr, 0 };
    emu_run_wave(dec_entry, &A);
    return (size_t)A.result;
}

#ifdef ZSA_EMU_ENCODE
struct EncArgs { ZsaEncShared* S; u32* ht; ZsaEncParams P; const u8* src; u64 srcSize; u8* dst; u64 cap; ZsaEncScratch W; const ZsaCDictDev* cd; int dictMode; u64 result; };
static void enc_entry(void* a)
{
    EncArgs* A = (EncArgs*)a;
    u64 r = (A->P.strat == 1 && !A->cd) ? zsa_encode_frame<2>(A->S, A->ht, A->P, A->src, A->srcSize, A->dst, A->cap, A->W, nullptr, 0)
          : (A->P.strat >= 2 && A->cd) ? zsa_encode_frame<4>(A->S, A->ht, A->P, A->src, A->srcSize, A->dst, A->cap, A->W, A->cd, A->dictMode)
          : A->P.strat >= 3 ? zsa_encode_frame<1>(A->S, A->ht, A->P, A->src, A->srcSize, A->dst, A->cap, A->W, nullptr, 0)
          : (A->P.strat == 2 && !A->cd) ? zsa_encode_frame<3>(A->S, A->ht, A->P, A->src, A->srcSize, A->dst, A->cap, A->W, nullptr, 0)
                            : zsa_encode_frame<0>(A->S, A->ht, A->P, A->src, A->srcSize, A->dst, A->cap, A->W, A->cd, A->dictMode);
    if (wv_lane() == 0) A->result = r;
}
extern "C" size_t emu_compress(void* dst, size_t cap, const void* src, size_t n, unsigned wlog, unsigned hlog, unsigned mml, unsigned tlen)
{
    static ZsaEncShared S;
    static u32* ht = (u32*)malloc(sizeof(u32) << 17);
    static u64* seq = (u64*)malloc(8 * (32768 + 64));
    static u8* lit = (u8*)malloc(ZSA_BLOCKSIZE_MAX + 128);
    static u8* tmp = (u8*)malloc(ZSA_BLOCKSIZE_MAX * 2 + 512);
    static const u8 empty[16] = {0};
    EncArgs A = { &S, ht, { wlog, hlog, mml, tlen, 0, 1 }, n ? (const u8*)src : empty + 8, n, (u8*)dst, cap, { seq, lit, tmp }, nullptr, 0, 0 };
    emu_run_wave(enc_entry, &A);
    return (size_t)A.result;
}

// parameters resolved from the level as the kernel does (ZSTD_fast .. ZSTD_lazy2 rows)
static u32* g_emu_ht;
extern "C" void* emu_last_tables() { return g_emu_ht; }   // the match tables as the last emu_compress_level call left them (debugging)
extern "C" size_t emu_compress_level(void* dst, size_t cap, const void* src, size_t n, int level)
{
    static ZsaEncShared S;
    static u32* ht = (u32*)malloc((size_t)9 << 20);
    g_emu_ht = ht;
    static u64* seq = (u64*)malloc(8 * (32768 + 64));
    static u8* lit = (u8*)malloc(ZSA_BLOCKSIZE_MAX + 128);
    static u8* tmp = (u8*)malloc(ZSA_BLOCKSIZE_MAX * 2 + 512);
    static ZsaFseCT park[3];
    static const u8 empty[16] = {0};
    ZsaCParams cp;
    if (zsa_resolveCParams(&cp, level, n, 0, zsa_cpm_noAttachDict)
        || ((size_t)4 << cp.hlog) + zsa_auxTableBytes(cp.strat, cp.wlog, cp.hlog, cp.clog) > ((size_t)9 << 20)) return (size_t)ZSA_ERR(ZSA_parameter_unsupported);
    EncArgs A = { &S, ht, { cp.wlog, cp.hlog, cp.mml, cp.tlen, cp.clog, cp.strat, cp.slog }, n ? (const u8*)src : empty + 8, n, (u8*)dst, cap, { seq, lit, tmp, park }, nullptr, 0, 0 };
    emu_run_wave(enc_entry, &A);
    return (size_t)A.result;
}

// ---- dictionaries: the digests the product builds on the device, built here by the same code on emulated lanes
struct EmuCDict { ZsaCDictDev cd; u8* dict; u32* table; };
struct EmuDDict { ZsaDDictDev dd; ZsaDDictTables T; u8* dict; };
struct DigArgs { ZsaEncShared* SE; ZsaDecShared* SD; EmuCDict* c; EmuDDict* d; u32 dictSize; ZsaCParams cp; int level; u32 result; };
static void digc_entry(void* a)
{
    DigArgs* A = (DigArgs*)a;
    u32 r = zsa_digest_cdict(A->SE, A->SD, A->c->dict, A->dictSize, A->cp, A->level, &A->c->cd, A->c->table, A->c->table + ((size_t)1 << A->cp.hlog));
    if (wv_lane() == 0) A->result = r;
}
static void digd_entry(void* a)
{
    DigArgs* A = (DigArgs*)a;
    u32 r = zsa_digest_ddict(A->SD, A->d->dict, A->dictSize, &A->d->dd, &A->d->T);
    if (wv_lane() == 0) A->result = r;
}
extern "C" void emu_free_cdict(void* h) { EmuCDict* c = (EmuCDict*)h; if (c) { free(c->dict); free(c->table); free(c); } }
extern "C" void emu_free_ddict(void* h) { EmuDDict* d = (EmuDDict*)h; if (d) { free(d->dict); free(d); } }
extern "C" void* emu_create_cdict(const void* dict, size_t n, int level)
{
    static ZsaEncShared SE; static ZsaDecShared SD;
    ZsaCParams cp;
    if (zsa_resolveCParams(&cp, level, ZSA_SIZE_UNKNOWN, n, zsa_cpm_createCDict)) return nullptr;
    EmuCDict* c = (EmuCDict*)calloc(1, sizeof(EmuCDict));
    c->dict = (u8*)malloc(n + 16); if (n) memcpy(c->dict, dict, n);
    c->table = (u32*)calloc(((size_t)1 << cp.hlog) + (cp.strat < 2 ? 0 : (zsa_useRowFinder(cp.strat, cp.wlog) ? (size_t)1 << cp.hlog : (size_t)1 << cp.clog)), 4);
    DigArgs A = { &SE, &SD, c, nullptr, (u32)n, cp, level, 0 };
    emu_run_wave(digc_entry, &A);
    if (A.result) { emu_free_cdict(c); return nullptr; }
    return c;
}
extern "C" void* emu_create_ddict(const void* dict, size_t n)
{
    static ZsaDecShared SD;
    EmuDDict* d = (EmuDDict*)calloc(1, sizeof(EmuDDict));
    d->dict = (u8*)malloc(n + 16); if (n) memcpy(d->dict, dict, n);
    DigArgs A = { nullptr, &SD, nullptr, d, (u32)n, {}, 0, 0 };
    emu_run_wave(digd_entry, &A);
    if (A.result) { emu_free_ddict(d); return nullptr; }
    return d;
}
extern "C" size_t emu_compress_cdict(void* dst, size_t cap, const void* src, size_t n, const void* h)
{
    static ZsaEncShared S;
    static u32* ht = (u32*)malloc(sizeof(u32) << 18);
    static u64* seq = (u64*)malloc(8 * (32768 + 64));
    static u8* lit = (u8*)malloc(ZSA_BLOCKSIZE_MAX + 128);
    static u8* tmp = (u8*)malloc(ZSA_BLOCKSIZE_MAX * 2 + 512);
    static const u8 empty[16] = {0};
    const EmuCDict* c = (const EmuCDict*)h;
    ZsaCParams cp; int mode = 0;
    if (zsa_resolveCDictUse(&cp, &mode, c->cd.cp, c->cd.level, c->cd.dictSize, n) || cp.hlog > 17 || cp.clog > 17
        || !(cp.strat == 1 || (mode == zsa_dm_attach && cp.strat == c->cd.cp.strat))) return (size_t)ZSA_ERR(ZSA_parameter_unsupported);
    EncArgs A = { &S, ht, { cp.wlog, cp.hlog, cp.mml, cp.tlen, cp.clog, cp.strat, cp.slog }, n ? (const u8*)src : empty + 8, n, (u8*)dst, cap, { seq, lit, tmp, nullptr }, &c->cd, mode, 0 };
    emu_run_wave(enc_entry, &A);
    return (size_t)A.result;
}
extern "C" size_t emu_decompress_ddict(void* dst, size_t cap, const void* src, size_t n, const void* h)
{
    static ZsaDecShared S;
    static u8* litBuf = (u8*)malloc(ZSA_DEC_WKSP_BYTES + 64);
    DecArgs A = { &S, (const u8*)src, n, (u8*)dst, cap, litBuf, &((const EmuDDict*)h)->dd, 0 };
    emu_run_wave(dec_entry, &A);
    return (size_t)A.result;
}

// ---- pieces of the entropy stage on their own, for tests against the reference's functions of the same name
// FSE_normalizeCount + FSE_writeNCount: count[0..maxSV] -> norm[0..maxSV], description bytes.  Returns the description's size,
// 0 when writing failed, (size_t)-1 when the normalisation failed.
struct NcArgs { ZsaEncShared* S; const unsigned* count; unsigned maxSV, total, tableLog, useLowProb; short* norm; u8* out; size_t result; };
static void nc_entry(void* a)
{
    NcArgs* A = (NcArgs*)a;
    int const lane = wv_lane();
    u32* const cnt = A->S->count;
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { cnt[i] = i <= A->maxSV ? A->count[i] : 0; A->S->norm[i] = 0; }
    wv_sync();
    u32 const bad = zsa_fse_normalizeCount_wave(A->S->norm, A->tableLog, cnt, A->total, A->maxSV, A->useLowProb, lane);
    wv_sync();
    size_t r = (size_t)-1;
    if (!bad) {
        if ((u32)lane <= A->maxSV) A->norm[lane] = A->S->norm[lane];
        r = zsa_fse_writeNCount_wave(A->S, A->out, A->S->norm, A->maxSV, A->tableLog, lane);
    }
    if (lane == 0) A->result = r;
}
extern "C" size_t emu_fse_ncount(const unsigned* count, unsigned maxSV, unsigned total, unsigned tableLog, unsigned useLowProb, short* norm, void* out)
{
    static ZsaEncShared S;
    NcArgs A = { &S, count, maxSV, total, tableLog, useLowProb, norm, (u8*)out, 0 };
    emu_run_wave(nc_entry, &A);
    return A.result;
}
// HUF_buildCTable_wksp: count[0..maxSV] -> code lengths and values per symbol.  Returns the table's maxNbBits (0: error).
struct HcArgs { ZsaEncShared* S; const unsigned* count; unsigned maxSV, maxNbBits; u8* nbBits; unsigned short* val; unsigned result; };
static void hc_entry(void* a)
{
    HcArgs* A = (HcArgs*)a;
    int const lane = wv_lane();
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) A->S->count[i] = i <= A->maxSV ? A->count[i] : 0;
    wv_sync();
    u32 const r = zsa_huf_buildCTable_wave(A->S, &A->S->hufNext, A->S->count, A->maxSV, A->maxNbBits, lane);
    wv_sync();
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { A->nbBits[i] = A->S->hufNext.nbBits[i]; A->val[i] = A->S->hufNext.val[i]; }
    if (lane == 0) A->result = r;
}
extern "C" unsigned emu_huf_ctable(const unsigned* count, unsigned maxSV, unsigned maxNbBits, unsigned char* nbBits, unsigned short* val)
{
    static ZsaEncShared S;
    HcArgs A = { &S, count, maxSV, maxNbBits, nbBits, val, 0 };
    emu_run_wave(hc_entry, &A);
    return A.result;
}
#endif
