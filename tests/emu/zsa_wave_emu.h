// zsa_wave_emu.h -- TEST-ONLY stand-in for zstd_amd/csrc/zsa_wave_hip.h.
//
// Lets tests compile the kernel BODIES (zsa_decode.h / zsa_encode.h) with g++ and run one workgroup
// (= one 64-lane wave) on the CPU: every lane is a ucontext fiber, and every wave primitive is a
// rendezvous of all 64 lanes.  This is a debugging aid for the HIP kernels (gdb, ASan/UBSan on the very
// same source); it is never part of libzstd_amd.so and proves nothing about GPU parity -- the `-m gpu`
// tests do that through the C-ABI.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <ucontext.h>

#define ZSA_DEV    static inline
#define ZSA_DEVFN  static
#define ZSA_MEM    inline
#define ZSA_HD     static inline
#define ZSA_SHARED static
#define ZSA_WAVE   64
#define ZSA_LDS

struct ZsaEmu {
    ucontext_t main_ctx;
    ucontext_t lane_ctx[64];
    char* stacks;
    int cur;                 // running lane
    int done[64];
    int ndone;
    // rendezvous
    uint64_t vals[2][64];
    int site[2];             // call-site line of the rendezvous in flight (divergence check)
    int arrived;
    unsigned gen;
};
extern ZsaEmu g_emu;

static inline void emu_yield()
{
    int from = g_emu.cur, next = from;
    for (int k = 0; k < 64; k++) { next = (next + 1) & 63; if (!g_emu.done[next]) break; }
    if (next == from) return;
    g_emu.cur = next;
    swapcontext(&g_emu.lane_ctx[from], &g_emu.lane_ctx[next]);
}
// all lanes contribute v; returns pointer to the 64 collected values (valid until the next-but-one rendezvous)
void emu_divergence(int lineA, int lineB, int lane);
static inline const uint64_t* emu_collect(uint64_t v, int line)
{
    unsigned mygen = g_emu.gen;
    uint64_t* slot = g_emu.vals[mygen & 1];
    // every wave primitive must be reached by all lanes from the same source line: catches divergent control flow
    if (g_emu.arrived == 0) g_emu.site[mygen & 1] = line;
    else if (g_emu.site[mygen & 1] != line) emu_divergence(g_emu.site[mygen & 1], line, g_emu.cur);
    slot[g_emu.cur] = v;
    if (++g_emu.arrived == 64 - g_emu.ndone) { g_emu.arrived = 0; g_emu.gen++; }
    else while (g_emu.gen == mygen) emu_yield();
    return slot;
}

ZSA_DEV int wv_lane() { return g_emu.cur; }
ZSA_DEV unsigned long long wv_ballot(bool p, int line = __builtin_LINE())
{
    const uint64_t* a = emu_collect(p ? 1 : 0, line);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) if (a[i]) m |= 1ull << i;
    return m;
}
ZSA_DEV unsigned wv_shfl(unsigned v, int srcLane, int line = __builtin_LINE()) { const uint64_t* a = emu_collect(v, line); return (unsigned)a[srcLane & 63]; }
ZSA_DEV unsigned wv_shfl_up(unsigned v, int delta, int line = __builtin_LINE()) { int me = g_emu.cur; const uint64_t* a = emu_collect(v, line); return me >= delta ? (unsigned)a[me - delta] : v; }
ZSA_DEV unsigned wv_shfl_xor(unsigned v, int mask, int line = __builtin_LINE()) { int me = g_emu.cur; const uint64_t* a = emu_collect(v, line); return (unsigned)a[(me ^ mask) & 63]; }
ZSA_DEV void wv_sync(int line = __builtin_LINE()) { (void)emu_collect(0, line); }
ZSA_DEV void wv_lds_fence(int line = __builtin_LINE()) { (void)emu_collect(0, line); }
ZSA_DEV void wv_wave_fence(int line = __builtin_LINE()) { (void)emu_collect(0, line); }
ZSA_DEV unsigned wv_readlane(unsigned v, int srcLane, int line = __builtin_LINE()) { const uint64_t* a = emu_collect(v, line); return (unsigned)a[srcLane & 63]; }
// wv_uniform: on the GPU a hint (readfirstlane); here a check that the value really is the same in all lanes
void emu_not_uniform(int line);
ZSA_DEV unsigned long long wv_uniform64(unsigned long long v, int line = __builtin_LINE())
{
    const uint64_t* a = emu_collect(v, line);
    for (int i = 0; i < 64; i++) if (!g_emu.done[i] && a[i] != v) emu_not_uniform(line);
    return v;
}
ZSA_DEV unsigned wv_uniform(unsigned v, int line = __builtin_LINE()) { return (unsigned)wv_uniform64(v, line); }
ZSA_DEV bool wv_uniform_b(bool v, int line = __builtin_LINE()) { return wv_uniform64(v ? 1 : 0, line) != 0; }
// a condition evaluated inside a section that ONE lane runs (no rendezvous: the other lanes are not there)
ZSA_DEV bool wv_single_lane_b(bool v) { return v; }
ZSA_DEV unsigned wv_single_lane_u32(unsigned v) { return v; }
ZSA_DEV unsigned wv_bfe(unsigned x, unsigned off, unsigned w) { w &= 31; return w ? (x >> (off & 31)) & ((1u << w) - 1u) : 0; }
ZSA_DEV unsigned wv_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3))); }
ZSA_DEV unsigned wv_ldg8(const unsigned char* p) { return *p; }
ZSA_DEV unsigned wv_ldg32(const unsigned char* p) { unsigned v; memcpy(&v, p, 4); return v; }
ZSA_DEV unsigned long long wv_ldg64(const unsigned char* p) { unsigned long long v; memcpy(&v, p, 8); return v; }
ZSA_DEV unsigned wv_vzero() { return 0; }
ZSA_DEV unsigned wv_ff1_64(unsigned long long x) { return x ? (unsigned)__builtin_ctzll(x) : 0xFFFFFFFFu; }
ZSA_DEV void wv_cold() {}
ZSA_DEV void wv_keep(unsigned) {}
ZSA_DEV unsigned wv_ldg_cell(const unsigned* p) { return *p; }
ZSA_DEV void wv_stg_cell(unsigned* p, unsigned v) { *p = v; }
ZSA_DEV void wv_stg64(void* p, unsigned long long v) { *(unsigned long long*)p = v; }
ZSA_DEV void wv_stg32(void* p, unsigned v) { memcpy(p, &v, 4); }
ZSA_DEV void wv_stg64u(void* p, unsigned long long v) { memcpy(p, &v, 8); }
ZSA_DEV void wv_stg16(void* p, unsigned v) { unsigned short w = (unsigned short)v; memcpy(p, &w, 2); }
ZSA_DEV void wv_stg8(unsigned char* p, unsigned v) { *p = (unsigned char)v; }
ZSA_DEV unsigned wv_ldg8_now(const unsigned char* p) { return wv_ldg8(p); }
ZSA_DEV unsigned wv_ldg32_now(const unsigned char* p) { return wv_ldg32(p); }
ZSA_DEV unsigned long long wv_ldg64_now(const unsigned char* p) { return wv_ldg64(p); }
template <typename A> ZSA_DEV void wv_join(A&, unsigned&, unsigned&) {}
template <typename A> ZSA_DEV void wv_join2(A&, unsigned&) {}
ZSA_DEV unsigned long long wv_clock() { return 0; }
ZSA_DEV unsigned wv_atomic_add_lds(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
ZSA_DEV unsigned wv_atomic_or_lds(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
ZSA_DEV unsigned wv_atomic_max_lds(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
ZSA_DEV unsigned wv_atomic_xor_lds(unsigned* p, unsigned v) { unsigned o = *p; *p = o ^ v; return o; }
ZSA_DEV unsigned wv_atomic_and_lds(unsigned* p, unsigned v) { unsigned o = *p; *p = o & v; return o; }
ZSA_DEV unsigned wv_atomic_add_global(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
ZSA_DEV void wv_atomic_max_global(unsigned* p, unsigned v) { if (v > *p) *p = v; }
ZSA_DEV void wv_device_fence() {}

ZSA_DEV unsigned wv_scan_incl(unsigned v, int line = __builtin_LINE())
{
    int me = g_emu.cur; const uint64_t* a = emu_collect(v, line);
    unsigned s = 0; for (int i = 0; i <= me; i++) s += (unsigned)a[i];
    return s;
}
ZSA_DEV unsigned wv_reduce_add(unsigned v, int line = __builtin_LINE()) { const uint64_t* a = emu_collect(v, line); unsigned s = 0; for (int i = 0; i < 64; i++) s += (unsigned)a[i]; return s; }
ZSA_DEV unsigned wv_reduce_max(unsigned v, int line = __builtin_LINE()) { const uint64_t* a = emu_collect(v, line); unsigned s = 0; for (int i = 0; i < 64; i++) if ((unsigned)a[i] > s) s = (unsigned)a[i]; return s; }

// run `fn(arg)` on 64 lanes
typedef void (*emu_kernel_fn)(void* arg);
void emu_run_wave(emu_kernel_fn fn, void* arg);
