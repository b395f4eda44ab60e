// zsa_wave_hip.h -- wave-level primitives for gfx950 (64-lane wavefronts).  The kernels are written against
// this small vocabulary only, so that tests can also run the same kernel bodies on the CPU under a
// lock-step lane emulator (tests/emu/zsa_wave_emu.h) to localise bugs before going to the GPU.
#pragma once
#include <hip/hip_runtime.h>

#define ZSA_DEV    __device__ __forceinline__
#define ZSA_DEVFN  __device__ __noinline__
#define ZSA_MEM    __device__ __forceinline__      // member functions
#define ZSA_HD     __host__ __device__ inline
#define ZSA_SHARED __shared__
#define ZSA_WAVE   64
// Qualifier of a pointer that is known to point into LDS: 32 bits wide, used by ds_* instructions as it is.  A generic pointer
// that differs from lane to lane (one table per lane, say) costs a 64-bit addition and a null check per access -- or a flat load.
#define ZSA_LDS    __attribute__((address_space(3)))

ZSA_DEV int wv_lane() { return (int)(threadIdx.x & 63u); }
ZSA_DEV unsigned long long wv_ballot(bool p) { return __ballot(p); }
ZSA_DEV unsigned wv_shfl(unsigned v, int srcLane) { return (unsigned)__shfl((int)v, srcLane, 64); }
ZSA_DEV unsigned wv_shfl_up(unsigned v, int delta) { return (unsigned)__shfl_up((int)v, (unsigned)delta, 64); }
ZSA_DEV unsigned wv_shfl_xor(unsigned v, int mask) { return (unsigned)__shfl_xor((int)v, mask, 64); }
// Workgroup == one wave here: orders LDS and global accesses between the lanes of the wave.
ZSA_DEV void wv_sync() { __syncthreads(); }
// One wave per workgroup: LDS operations of a wave are issued and completed in order, so making one lane's LDS write
// visible to another lane's later LDS read needs no s_barrier and no wait on outstanding global loads -- only a
// compiler-level ordering point.
ZSA_DEV void wv_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// The same ordering point for GLOBAL memory handed from lane to lane inside one wave (a lane's earlier stores, flat or
// address-space-1, against another lane's later loads of the same bytes: the decoder's dependent matches, the parsers'
// match tables).  Sound for a single wave under the AMDGPU memory model: a wave's memory instructions are issued in program
// order and a wavefront-scope fence keeps the compiler from moving them across it; nothing here synchronises two waves.
ZSA_DEV void wv_wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// value of `v` in lane `srcLane`; srcLane must be wave-uniform
ZSA_DEV unsigned wv_readlane(unsigned v, int srcLane) { return (unsigned)__builtin_amdgcn_readlane((int)v, srcLane); }
// Tells the compiler that `v` holds the same value in every lane (it cannot always prove it: a value that once passed through
// per-lane control flow stays "divergent" for it, and a divergent parse position turns every branch of the parse loop into
// exec-mask bookkeeping).  The result lives in a scalar register; everything computed from it is scalar work.
ZSA_DEV unsigned wv_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
ZSA_DEV unsigned long long wv_uniform64(unsigned long long v)
{ return (unsigned long long)wv_uniform((unsigned)v) | ((unsigned long long)wv_uniform((unsigned)(v >> 32)) << 32); }
ZSA_DEV bool wv_uniform_b(bool v) { return wv_uniform(v ? 1u : 0u) != 0; }
// A condition evaluated inside a section that ONE lane runs (`if (lane == 0) { ... }`): for the compiler the section is divergent
// code and every branch in it becomes exec-mask bookkeeping (save, xor, restore: five scalar instructions); with a single lane
// in the mask the first active lane IS the lane, so its condition can be the condition of a scalar branch.
ZSA_DEV bool wv_single_lane_b(bool v) { return __ballot(v) != 0; }   // (the compare writes the mask, the branch tests it: v_cmp, s_cmp_lg_u64, s_cbranch)
// ... and a value computed there can be a scalar register (loop bounds: the counter then stays on the scalar unit)
ZSA_DEV unsigned wv_single_lane_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// bytes [sh, sh+4) of the little-endian 8-byte value hi:lo, sh in 0..3
// bits [off, off + w) of x (w == 0 gives 0; off is taken modulo 32)
ZSA_DEV unsigned wv_bfe(unsigned x, unsigned off, unsigned w) { return __builtin_amdgcn_ubfe(x, off, w); }
ZSA_DEV unsigned wv_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
// loads from a pointer known to be global memory (frame descriptors hand out generic pointers: without this the compiler
// emits flat_load, which also ties up the LDS counter); unaligned addresses are fine on gfx950
struct __attribute__((packed)) WvPk32 { unsigned v; };
struct __attribute__((packed)) WvPk64 { unsigned long long v; };
ZSA_DEV unsigned wv_ldg8(const unsigned char* p) { return *(const __attribute__((address_space(1))) unsigned char*)p; }
ZSA_DEV unsigned wv_ldg32(const unsigned char* p) { return ((const __attribute__((address_space(1))) WvPk32*)p)->v; }
ZSA_DEV unsigned long long wv_ldg64(const unsigned char* p) { return ((const __attribute__((address_space(1))) WvPk64*)p)->v; }
// A zero the compiler must treat as a per-lane (vector) value.  Serial sections run by one lane derive all their state from
// uniform inputs, so the compiler scalarises them -- and a scalarised chain has to pull every loaded value into SGPRs
// (v_readfirstlane) right behind its load, i.e. wait for it at once.  Seeding such a section's position with this zero
// keeps its state in VGPRs, where loads can stay in flight until their first real use.
// index of the lowest set bit of a wave-uniform 64-bit value, 0xFFFFFFFF for zero: s_ff1_i32_b64 as it is (the compiler's own
// lowering of "ctz, or a constant for zero" adds a compare and a select)
ZSA_DEV unsigned wv_ff1_64(unsigned long long x) { unsigned r; asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(x)); return r; }
// Keeps the block it stands in a block of its own: the compiler may not turn the branch around it into selects that every
// pass of the enclosing loop would then pay for (used on loop exits that carry bookkeeping).
ZSA_DEV void wv_cold() { asm volatile(""); }
// keeps a loaded value alive up to this point (a load made for its effect on the cache)
ZSA_DEV void wv_keep(unsigned v) { asm volatile("" :: "v"(v)); }
ZSA_DEV unsigned wv_vzero() { unsigned z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }
// aligned table cells in global memory
ZSA_DEV unsigned wv_ldg_cell(const unsigned* p) { return *(const __attribute__((address_space(1))) unsigned*)p; }
ZSA_DEV void wv_stg_cell(unsigned* p, unsigned v) { *(__attribute__((address_space(1))) unsigned*)p = v; }
ZSA_DEV void wv_stg64(void* p, unsigned long long v) { *(__attribute__((address_space(1))) unsigned long long*)p = v; }
ZSA_DEV void wv_stg32(void* p, unsigned v) { ((__attribute__((address_space(1))) WvPk32*)p)->v = v; }
ZSA_DEV void wv_stg64u(void* p, unsigned long long v) { ((__attribute__((address_space(1))) WvPk64*)p)->v = v; }   // unaligned
struct __attribute__((packed)) WvPk16 { unsigned short v; };
ZSA_DEV void wv_stg16(void* p, unsigned v) { ((__attribute__((address_space(1))) WvPk16*)p)->v = (unsigned short)v; }   // unaligned
ZSA_DEV void wv_stg8(unsigned char* p, unsigned v) { *(__attribute__((address_space(1))) unsigned char*)p = (unsigned char)v; }
// The same loads, completed before control leaves the call.  For the rarely-taken global side of an "LDS or global"
// read: gfx950 counts loads and stores in one counter (vmcnt), so if a pending load could reach the join of the two
// sides the compiler must put `s_waitcnt vmcnt(0)` after the join -- and on the common LDS side that wait then stalls on
// the acknowledgement of whatever stores are still in flight (a full L2 round trip in the middle of the parse chain).
ZSA_DEV unsigned wv_ldg8_now(const unsigned char* p) { unsigned v = wv_ldg8(p); asm volatile("" : "+v"(v)); return v; }
ZSA_DEV unsigned wv_ldg32_now(const unsigned char* p) { unsigned v = wv_ldg32(p); asm volatile("" : "+v"(v)); return v; }
ZSA_DEV unsigned long long wv_ldg64_now(const unsigned char* p) { unsigned long long v = wv_ldg64(p); asm volatile("" : "+v"(v)); return v; }
// Loads issued together, consumed together: pins the point where all of the given loaded values must have arrived, so the
// compiler cannot sink each load into the branch that uses it (which would turn one memory round trip into several).
template <typename A> ZSA_DEV void wv_join(A& a, unsigned& b, unsigned& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
template <typename A> ZSA_DEV void wv_join2(A& a, unsigned& b) { asm volatile("" : "+v"(a), "+v"(b)); }
ZSA_DEV unsigned long long wv_clock() { return (unsigned long long)clock64(); }
ZSA_DEV unsigned wv_atomic_add_lds(unsigned* p, unsigned v) { return atomicAdd(p, v); }
ZSA_DEV unsigned wv_atomic_or_lds(unsigned* p, unsigned v) { return atomicOr(p, v); }
ZSA_DEV unsigned wv_atomic_max_lds(unsigned* p, unsigned v) { return atomicMax(p, v); }
ZSA_DEV unsigned wv_atomic_xor_lds(unsigned* p, unsigned v) { return atomicXor(p, v); }
ZSA_DEV unsigned wv_atomic_and_lds(unsigned* p, unsigned v) { return atomicAnd(p, v); }
ZSA_DEV unsigned wv_atomic_add_global(unsigned* p, unsigned v) { return atomicAdd(p, v); }
ZSA_DEV void wv_atomic_max_global(unsigned* p, unsigned v) { (void)atomicMax(p, v); }
// device-scope fence: afterwards plain loads of this wave observe earlier device-scope atomics (they bypass the L1)
ZSA_DEV void wv_device_fence() { __threadfence(); }

// Inclusive prefix sum / sum / maximum over the 64 lanes on the data-parallel-primitive path of the vector unit: four row
// shifts inside each row of 16 lanes, then the last lane of row 0 (2) broadcast into row 1 (3) and the last lane of row 1 into
// rows 2 and 3 (GFX9's row_bcast:15 / row_bcast:31) -- six VALU operations without a trip through the LDS crossbar (the
// shuffle-based forms they replace were six ds_bpermute round trips each).  Lanes without a source keep the identity (0).
// All 64 lanes must be active.
#define ZSA_DPP_STEP(op, v, ctrl, rowMask) v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rowMask, 0xf, false))
#define ZSA_DPP_SCAN(op, v) do { ZSA_DPP_STEP(op, v, 0x111, 0xf); ZSA_DPP_STEP(op, v, 0x112, 0xf); ZSA_DPP_STEP(op, v, 0x114, 0xf); \
                                 ZSA_DPP_STEP(op, v, 0x118, 0xf); ZSA_DPP_STEP(op, v, 0x142, 0xa); ZSA_DPP_STEP(op, v, 0x143, 0xc); } while (0)
ZSA_DEV unsigned wv_op_add(unsigned a, unsigned b) { return a + b; }
ZSA_DEV unsigned wv_op_max(unsigned a, unsigned b) { return a > b ? a : b; }
ZSA_DEV unsigned wv_scan_incl(unsigned v) { ZSA_DPP_SCAN(wv_op_add, v); return v; }
ZSA_DEV unsigned wv_reduce_add(unsigned v) { ZSA_DPP_SCAN(wv_op_add, v); return wv_readlane(v, 63); }
ZSA_DEV unsigned wv_reduce_max(unsigned v) { ZSA_DPP_SCAN(wv_op_max, v); return wv_readlane(v, 63); }
