// emu_driver.cpp -- TEST-ONLY: runs the HIP kernel bodies on the CPU, one emulated 64-lane wave at a time.
// Built by tests/emu/Makefile into tests/emu/libzsa_emu.so and driven from tests/test_emu_*.py.
// Never part of the product; the product library has no CPU path.
#define ZSA_WAVE_HEADER "zsa_wave_emu.h"
#include "zsa_common.h"
#include "zsa_decode.h"
#ifdef ZSA_EMU_ENCODE
#include "zsa_encode.h"
#endif
#include <stdio.h>

ZsaEmu g_emu;
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
    static u8* litBuf = (u8*)malloc(ZSA_BLOCKSIZE_MAX + 64);
    DecArgs A = { &S, (const u8*)src, n, (u8*)dst, cap, litBuf, nullptr, 0 };
    emu_run_wave(dec_entry, &A);
    return (size_t)A.result;
}

#ifdef ZSA_EMU_ENCODE
struct EncArgs { ZsaEncShared* S; u32* ht; ZsaEncParams P; const u8* src; u64 srcSize; u8* dst; u64 cap; ZsaEncScratch W; const ZsaCDictDev* cd; u64 result; };
static void enc_entry(void* a)
{
    EncArgs* A = (EncArgs*)a;
    u64 r = zsa_encode_frame(A->S, A->ht, A->P, A->src, A->srcSize, A->dst, A->cap, A->W, A->cd);
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
    EncArgs A = { &S, ht, { wlog, hlog, mml, tlen }, n ? (const u8*)src : empty + 8, n, (u8*)dst, cap, { seq, lit, tmp }, nullptr, 0 };
    emu_run_wave(enc_entry, &A);
    return (size_t)A.result;
}
#endif
