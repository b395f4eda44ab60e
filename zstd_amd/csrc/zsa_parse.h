// zsa_parse.h -- the ZSTD_fast parse (zstd_fast.c:143-403, no dictionary) executed by all 64 lanes of the wave.
//
// The reference walks positions in pairs (ip0, ip0+1), then (ip0+step, ip0+step+1), ... and per pair performs, in
// this order: a repcode check at the NEXT pair's first position, a hash-table check at ip0, a hash-table check at
// ip0+1; every visited position also replaces its hash-table cell ("read the cell, then write my position", in
// visiting order).  That is one serial chain, but everything between two matches can be evaluated speculatively:
//
//   * a WINDOW is ZSA_NP consecutive pairs.  Lane 3i evaluates the repcode check of pair i, lanes 3i+1 and 3i+2 the
//     two hash checks -- lane order == the reference's event order, so the first set bit of one ballot is the event
//     the reference would have stopped at;
//   * the hash table lives in global memory (L2-resident, 4 << hashLog bytes per resident workgroup): keeping it out of
//     LDS is what lets 8 workgroups share a CU, and the table gathers of one wave overlap with the work of the others;
//   * all hash lanes read their cell; whether two lanes of the window share a cell is detected with a one-byte tag
//     array in LDS (write my lane id at tag[h], read it back).  If not (the common case) the cell values are exactly
//     what the serial walk would have seen.  Otherwise (runs of equal bytes) the nearest earlier lane with the same
//     cell is found with a uniform readlane loop;
//   * once the winning event is known, the lanes up to it store their positions (per cell, the last one in window
//     order: positions grow with lane order);
//   * match extension compares 64 x 8 bytes per step forwards and 64 bytes per step backwards, one ballot each.
//
// Source bytes come from an LDS ring that trails the parse position (filled 1 KiB per step with 16-byte coalesced loads),
// so the per-sequence dependency chain is LDS latency except for match candidates older than the ring, which come
// from L2.  Sequences are staged in LDS and written out 64 at a time.
//
// Level 1 / 2 frames without a dictionary take the CHUNK path (ZsaTabFilt, the block at the top of the window loop): the
// per-position work that does not depend on the parse state -- source word, hash, fingerprint, probe -- is done once for
// 64 consecutive positions, and the probe does not touch the global table at all: an LDS array holds the top 4 fingerprint
// bits of every cell, so only the position that wins a search reads its cell (one 4-byte load per match instead of a
// 64-cell gather per window).  The general window evaluation remains as the path for everything the chunk path leaves
// out (step > 2 after 128 bytes without a match, the last pairs of a block, table cells shared by neighbouring lanes).
// Included by zsa_encode.h (uses zsa_hash / zsa_packseq / ZsaEncParams from there).
#pragma once

// the ring is followed by 16 bytes that mirror its first 16, so unaligned reads may run over the end
#ifndef ZSA_NP
#define ZSA_NP 7           // pairs per window: 21 lanes (matches come every ~3 positions on text-like data)
#endif

#ifdef ZSA_PARSE_PROF
#define ZSA_PPROF(S, lane, slot) ZSA_PROF(S, lane, slot)
#else
#define ZSA_PPROF(S, lane, slot) do {} while (0)
#endif
#ifndef ZSA_STAT          // test builds count how the hoisted window fares (tests/emu)
#define ZSA_STAT(i) do {} while (0)
#endif
struct ZsaEncShared;

struct ZsaRing {
    u8* b;                 // LDS, ZSA_RING_BYTES + 16 bytes; byte at frame offset o lives at b[o & (ZSA_RING_BYTES-1)]
    const u8* src;         // frame start
    u32 lo, hi;            // ring holds frame offsets [lo, hi)
    u32 limit;             // no byte at or beyond this frame offset is ever read (end of the block being parsed)
    u32 min;               // ... nor below this one (a frame that follows a dictionary in index space starts above 0)
};
struct __attribute__((packed)) ZsaPk64 { u64 v; };
struct __attribute__((packed)) ZsaPk32 { u32 v; };

// ring-only reads (gfx950 LDS handles unaligned ds_read_b32/b64): the caller guarantees residency
ZSA_DEV u64 zsa_ring64(const ZsaRing& R, u32 o) { return ((const ZsaPk64*)(R.b + (o & (ZSA_RING_BYTES - 1))))->v; }
ZSA_DEV u32 zsa_ring32(const ZsaRing& R, u32 o) { return ((const ZsaPk32*)(R.b + (o & (ZSA_RING_BYTES - 1))))->v; }
ZSA_DEV u32 zsa_ring8(const ZsaRing& R, u32 o) { return R.b[o & (ZSA_RING_BYTES - 1)]; }
// reads that fall back to global memory (L2) for bytes the ring no longer holds
ZSA_DEV u32 zsa_rd32(const ZsaRing& R, u32 o) { return (o >= R.lo && o + 4 <= R.hi) ? zsa_ring32(R, o) : wv_ldg32_now(R.src + o); }
ZSA_DEV u64 zsa_rd64(const ZsaRing& R, u32 o) { return (o >= R.lo && o + 8 <= R.hi) ? zsa_ring64(R, o) : wv_ldg64_now(R.src + o); }
ZSA_DEV u32 zsa_rd8(const ZsaRing& R, u32 o) { return (o >= R.lo && o < R.hi) ? zsa_ring8(R, o) : wv_ldg8_now(R.src + o); }

// make [.., min(want, limit)) resident; `want` is uniform
ZSA_DEV void zsa_ring_fill(ZsaRing& R, u32 want, int lane)
{
    wv_lds_fence();   // every lane is done reading the slots that are about to be replaced
    while (R.hi < want && R.hi < R.limit) {
        u32 const o = R.hi + 16u * (u32)lane;
        u32 v[4] = { 0, 0, 0, 0 };
        if (o + 16 <= R.limit && o >= R.min) { v[0] = wv_ldg32(R.src + o); v[1] = wv_ldg32(R.src + o + 4); v[2] = wv_ldg32(R.src + o + 8); v[3] = wv_ldg32(R.src + o + 12); }
        else if (o < R.limit && o + 16 > R.min) { for (u32 k = 0; k < 16; k++) if (o + k >= R.min && o + k < R.limit) v[k >> 2] |= wv_ldg8(R.src + o + k) << (8 * (k & 3)); }
        u32 const ro = o & (ZSA_RING_BYTES - 1);
        u32* const d = (u32*)(R.b + ro);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        if (ro == 0) { u32* const m = (u32*)(R.b + ZSA_RING_BYTES); m[0] = v[0]; m[1] = v[1]; m[2] = v[2]; m[3] = v[3]; }
        R.hi += 1024;
    }
    if (R.hi - R.lo > ZSA_RING_BYTES) R.lo = R.hi - ZSA_RING_BYTES;
    wv_lds_fence();
}

// common-prefix length of [ip, iend) and [m, ...), m < ip (ZSTD_count, zstd_compress_internal.h:755-774), all lanes;
// arguments are frame offsets and uniform
ZSA_DEV u32 zsa_count_wave(const ZsaRing& R, u32 ip, u32 m, u32 iend, int lane)
{
    u32 total = 0;
    for (;;) {
        u32 const a = ip + total + 8u * (u32)lane, b = m + total + 8u * (u32)lane;
        u32 eq;
        if (ip + total + 512 <= iend && ip + total + 512 <= R.hi && m + total >= R.lo) {   // uniform: all 2 x 512 bytes in the ring
            u64 const diff = zsa_ring64(R, a) ^ zsa_ring64(R, b);
            eq = diff ? ((u32)__builtin_ctzll(diff) >> 3) : 8;
        } else if (a + 8 <= iend) {
            u64 const diff = zsa_rd64(R, a) ^ zsa_rd64(R, b);
            eq = diff ? ((u32)__builtin_ctzll(diff) >> 3) : 8;
        } else {
            u32 const n = a < iend ? iend - a : 0;
            eq = 0;
            while (eq < n && zsa_rd8(R, a + eq) == zsa_rd8(R, b + eq)) eq++;
        }
        u64 const stop = wv_ballot(eq < 8);
        if (stop) {
            int const f = __builtin_ctzll(stop);
            return total + 8u * (u32)f + wv_readlane(eq, f);
        }
        total += 512;
    }
}

// One round of match extension: lanes 0..47 compare 8 bytes each forwards from (ip, m) (capped at iend), lanes 48..63
// one byte each backwards from (ipB-1, mB-1) for at most maxBack bytes -- both directions ride on the same memory
// round trip.  Returns forward length in *fwd and backward length in *bwd (both complete: longer runs loop).
ZSA_DEV void zsa_extend_wave(const ZsaRing& R, u32 ip, u32 m, u32 iend, u32 ipB, u32 mB, u32 maxBack, u32* fwd, u32* bwd, int lane)
{
    bool stop;
    u32 eq = 0;
    if (ip + 384 <= iend && ip + 384 <= R.hi && mB >= R.lo + 16) {
        // uniform common case: both compare regions lie in the ring and clear of the block end -- no per-lane paths.
        // (ring addresses are masked, so the reads of lanes that do not take part are harmless)
        bool const fw = lane < 48;
        u32 const k = (u32)lane - 48u;
        u32 const a = fw ? ip + 8u * (u32)lane : ipB - 1 - k, b = fw ? m + 8u * (u32)lane : mB - 1 - k;
        u64 const diff = zsa_ring64(R, a) ^ zsa_ring64(R, b);
        eq = diff ? ((u32)__builtin_ctzll(diff) >> 3) : 8;
        stop = fw ? eq < 8 : !((k < maxBack) && eq > 0);
    } else if (lane < 48) {
        u32 const a = ip + 8u * (u32)lane, b = m + 8u * (u32)lane;
        if (a + 8 <= iend) {
            u64 const diff = zsa_rd64(R, a) ^ zsa_rd64(R, b);
            eq = diff ? ((u32)__builtin_ctzll(diff) >> 3) : 8;
        } else {
            u32 const n = a < iend ? iend - a : 0;
            while (eq < n && zsa_rd8(R, a + eq) == zsa_rd8(R, b + eq)) eq++;
        }
        stop = eq < 8;
    } else {
        u32 const k = (u32)lane - 48u;
        stop = !((k < maxBack) && (zsa_rd8(R, ipB - 1 - k) == zsa_rd8(R, mB - 1 - k)));
    }
    u64 const sb = wv_ballot(stop);
    u64 const sf = sb & 0xFFFFFFFFFFFFull, sk = sb >> 48;
    if (sf) { int const f = __builtin_ctzll(sf); *fwd = 8u * (u32)f + wv_readlane(eq, f); }
    else *fwd = 384 + zsa_count_wave(R, ip + 384, m + 384, iend, lane);
    if (sk) *bwd = (u32)__builtin_ctzll(sk);
    else {
        u32 nb = 16;
        while (nb < maxBack) {   // zstd_fast.c:285-289, 64 bytes per step
            u32 const k = nb + (u32)lane;
            bool const e = (k < maxBack) && (zsa_rd8(R, ipB - 1 - k) == zsa_rd8(R, mB - 1 - k));
            u64 const ne = wv_ballot(!e);
            u32 const f = ne ? (u32)__builtin_ctzll(ne) : 64u;
            nb += f;
            if (f < 64) break;
        }
        *bwd = nb;
    }
}

// ZSTD_hashPtr for a compile-time minMatch (zstd_compress_internal.h:809-846)
template <int MLS> ZSA_DEV u32 zsa_hashT(u64 v, u32 hlog)
{
    if (MLS <= 4) return ((u32)v * 2654435761U) >> (32 - hlog);
    if (MLS == 5) return (u32)(((v << 24) * 889523592379ULL) >> (64 - hlog));
    if (MLS == 6) return (u32)(((v << 16) * 227718039650203ULL) >> (64 - hlog));
    if (MLS == 7) return (u32)(((v << 8) * 58295818150454627ULL) >> (64 - hlog));
    return (u32)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - hlog));
}

// Match-table accessors.  A cell is 32 bits: the low posBits bits hold the position, the bits above them a fingerprint of
// the 4 bytes at that position (or nothing, posBits == 0: plain 32-bit positions as in the reference).
struct ZsaTabGlobal {   // one 32-bit word per cell in the workgroup's global scratch
    u32* p;
    static constexpr bool filtered = false;
    ZSA_MEM u32 ld(u32 h) const { return p[h]; }
    ZSA_MEM void st(u32 h, u32 c) const { p[h] = c; }
    ZSA_MEM u32 fget(u32) const { return 0; }
    ZSA_MEM void fxor(u32, u32) const {}
    ZSA_MEM u32* dupWord(u32) const { return nullptr; }
};
// The same cells plus, in LDS, a FILTER: the top 4 bits of every cell (fingerprint bits), 8 cells per word, kept equal to
// the global cell by every store.  `dup` is a bitmap of 1 << hashLog bits, all zero between uses (chunk set-up).
struct ZsaTabFilt {
    u32* p; u32* filt; u32* dup;
    u32 fb;            // filter bits per cell: 4 (tables of up to 8192 cells) or 2 (16384 cells) -- the filter is 4 KiB either way
    static constexpr bool filtered = true;
    ZSA_MEM u32 ld(u32 h) const { return wv_ldg_cell(p + h); }
    ZSA_MEM u32 top(u32 cell) const { return cell >> (32u - fb); }                 // the bits of a cell that the filter mirrors
    ZSA_MEM u32 widx(u32 h) const { return fb == 4u ? h >> 3 : h >> 4; }
    ZSA_MEM u32 wsh(u32 h) const { return fb == 4u ? (h & 7u) * 4u : (h & 15u) * 2u; }
    ZSA_MEM u32 fget(u32 h) const { return (filt[widx(h)] >> wsh(h)) & ((1u << fb) - 1u); }
    // d = old bits ^ new bits.  An atomic xor: lanes that update different cells of one word do not disturb each other
    ZSA_MEM void fxor(u32 h, u32 d) const { (void)wv_atomic_xor_lds(&filt[widx(h)], d << wsh(h)); }
    ZSA_MEM void st(u32 h, u32 c) const { wv_stg_cell(p + h, c); u32 const d = fget(h) ^ top(c); if (d) fxor(h, d); }
    ZSA_MEM u32* dupWord(u32 h) const { return &dup[(h >> 5) & 255u]; }             // (cell numbers modulo 8192: a false share only cuts a chunk short)
};

// One round of match extension for the chunk path: lanes 0..15 compare 8 bytes each forwards from (ip, m) (capped at
// iend), lanes 48..63 one byte each backwards from (ipB-1, mB-1) for at most maxBack bytes; 128 + 16 bytes per side keep a
// candidate that is older than the ring to one or two cache lines.  Longer runs continue in the wide loops.
ZSA_DEV void zsa_extend16(const ZsaRing& R, u32 ip, u32 m, u32 iend, u32 ipB, u32 mB, u32 maxBack, u32* fwd, u32* bwd, int lane)
{
    if (!(ip + 136 <= iend && ip + 136 <= R.hi && ipB >= R.lo + 16)) { zsa_extend_wave(R, ip, m, iend, ipB, mB, maxBack, fwd, bwd, lane); return; }
    bool const fw = lane < 16;
    u32 const k = (u32)lane - 48u;
    u32 const a = fw ? ip + 8u * (u32)lane : ipB - 1 - k, b = fw ? m + 8u * (u32)lane : mB - 1 - k;
    u64 const sa = zsa_ring64(R, a);                                 // masked ring address: legal in every lane
    u64 sb;
    if (mB >= R.lo + 16) sb = zsa_ring64(R, b);                      // uniform: the candidate lies in the ring as well
    else { sb = ~sa; if (fw || (lane >= 48 && k < maxBack)) sb = wv_ldg64_now(R.src + b); }
    u64 const diff = sa ^ sb;
    u32 const eq = diff ? ((u32)__builtin_ctzll(diff) >> 3) : 8;
    bool const stop = fw ? eq < 8 : !((k < maxBack) && eq > 0);
    u64 const sbal = wv_ballot(stop);
    u64 const sf = sbal & 0xFFFFull, sk = sbal >> 48;
    if (sf) { int const f = __builtin_ctzll(sf); *fwd = 8u * (u32)f + wv_readlane(eq, f); }
    else *fwd = 128 + zsa_count_wave(R, ip + 128, m + 128, iend, lane);
    if (sk) *bwd = (u32)__builtin_ctzll(sk);
    else {
        u32 nb = 16;
        while (nb < maxBack) {   // zstd_fast.c:285-289, 64 bytes per step
            u32 const kk = nb + (u32)lane;
            bool const e = (kk < maxBack) && (zsa_rd8(R, ipB - 1 - kk) == zsa_rd8(R, mB - 1 - kk));
            u64 const ne = wv_ballot(!e);
            u32 const f = ne ? (u32)__builtin_ctzll(ne) : 64u;
            nb += f;
            if (f < 64) break;
        }
        *bwd = nb;
    }
}

// ---- what a block's parse carries from one function to the next (all wave-uniform)
struct ZsaParseCtx {      // constants of the block
    const u8* base; u64* seqOut;
    u32 hlog, posBits, stepSize, iend, ilimit, prefixStartIndex;
    u32 ringMin;          // lowest frame offset the source ring may load (see ZsaRing::min)
};
struct ZsaParseState {
    u32 ip0, anchor, rep1, rep2, nbSeq;
    u32 chkRep2;          // the immediate-repcode check at ip0 (zstd_fast.c:311-324) is still to be made
    u32 nextStep;         // midEpisode: the search that is under way raises its step at this position (:322-327)
    u32 midEpisode;
    u32 rlo, rhi;         // what the source ring holds
    u32 done;             // the block is parsed
};
#ifdef ZSA_FP_TEST_BITS   // tests: a few fingerprint bits only, so that collisions are common
#define ZSA_FPV(x) ((((u32)(x) * 2246822519U) >> posBits) & ((1u << ZSA_FP_TEST_BITS) - 1u))
#else
#define ZSA_FPV(x) (((u32)(x) * 2246822519U) >> posBits)
#endif
#define ZSA_CELL(p, x) (useFp ? ((ZSA_FPV(x) << posBits) | (p)) : (p))
#define RD32(i) zsa_rd32(R, (i) - 2)
#define RD64(i) zsa_rd64(R, (i) - 2)
#define RD8(i)  zsa_rd8(R, (i) - 2)

// The general window evaluation.  Runs from st.ip0 to the end of the block, or -- `until` != 0xFFFFFFFF -- hands back at the
// first search restart (label _start of the reference) at or beyond position `until`.
// Table cells.  posBits == 0: a cell is the position, as in the reference.  posBits > 0 (frames without a dictionary,
// whose table only this parser touches): the low posBits bits are the position and the bits above them a fingerprint
// of the 4 bytes at that position.  A probe whose fingerprint differs from the current 4 bytes cannot match, so only
// the event that wins the window ever reads its candidate's bytes -- and that read is the first round of the match
// extension, which then starts at byte 0 and doubles as the 4-byte check the reference makes (a fingerprint collision
// shows up as a common prefix shorter than 4 and the next event of the window is taken instead).
template <int MLS, class TAB>
ZSA_DEV void zsa_parse_general_body(const TAB T, u32* ringWords, u8* tags, const ZsaParseCtx& C, ZsaParseState& st, u32 until, int lane)
{
    u32 const hlog = C.hlog, posBits = C.posBits, stepSize = C.stepSize, iend = C.iend, ilimit = C.ilimit, prefixStartIndex = C.prefixStartIndex;
    u64* const seqOut = C.seqOut;
    bool const useFp = posBits != 0;
    u32 const posMask = useFp ? ((1u << posBits) - 1u) : 0xFFFFFFFFu;
    ZsaRing R;
    R.b = (u8*)ringWords; R.src = C.base + 2; R.limit = iend - 2; R.min = C.ringMin; R.lo = wv_uniform(st.rlo); R.hi = wv_uniform(st.rhi);
    u32 anchor = wv_uniform(st.anchor), ip0 = wv_uniform(st.ip0);
    u32 rep1 = wv_uniform(st.rep1), rep2 = wv_uniform(st.rep2), nbSeq = wv_uniform(st.nbSeq);
    bool resume = wv_uniform(st.midEpisode) != 0;
    u32 const resumeNext = wv_uniform(st.nextStep);
    u32 const pair = (u32)lane / 3u, role = (u32)lane - 3u * pair;   // role 0: repcode check, 1: hash ip0, 2: hash ip0+1
    bool const inWin = lane < 3 * ZSA_NP;
    bool const hashLane = inWin && role != 0;
    // with step 2 (level 1 until 128 bytes pass without a match) the lane's position is ip0 + a constant
    u32 const fastOff = role == 0 ? 2u * pair + 2u : 2u * pair + (role == 2);

    // The immediate-repcode check that follows every match (zstd_fast.c:311-324) rides along with the next window as one
    // more event (lane 63) that precedes all others: its compare shares the window's memory round trip.
    bool chkRep2 = wv_uniform(st.chkRep2) != 0;
    // The source words of the first window after a match are requested as soon as the match end is known, so that their
    // LDS round trip overlaps with the post-match table inserts instead of following them.
    u64 vPre = 0; bool havePre = false;
    bool blockDone = false, first = true;
    for (;;) {   // one search episode per iteration (label _start)
        if (!first && ip0 >= until) break;                       // hand back at a search restart (after at least one search)
        first = false;
        u32 step = stepSize, nextStep = ip0 + 128, sNext = stepSize;
        if (resume) { nextStep = resumeNext; resume = false; }
        if (ip0 + step + 1 >= ilimit) {
            // no room for a window: finish the immediate-repcode loop on its own
            while (chkRep2 && ip0 <= ilimit && zsa_ring32(R, ip0 - 2) == RD32(ip0 - rep2)) {
                u32 const rLength = zsa_count_wave(R, ip0 + 4 - 2, ip0 + 4 - rep2 - 2, iend - 2, lane) + 4;
                { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                {   u64 const wd = zsa_ring64(R, ip0 - 2);
                    u32 const hh = zsa_hashT<MLS>(wd, hlog);
                    if (lane == 0) T.st(hh, ZSA_CELL(ip0, wd));
                    wv_lds_fence();
                }
                ip0 += rLength;
                if (lane == 0) seqOut[nbSeq] = zsa_packseq(0, rLength - ZSA_MINMATCH, 1);
                ZSA_STAT(5);
                nbSeq++;
                anchor = ip0;
                if (R.hi < ip0 - 2 + 512 && R.hi < R.limit) zsa_ring_fill(R, ip0 - 2 + 1024, lane);
            }
            blockDone = true;
            break;
        }
        for (;;) {   // one window per iteration
            {   u32 const want = ip0 - 2 + 1024;
                if (R.hi < want && R.hi < R.limit) zsa_ring_fill(R, want, lane);
            }

            // ---- positions of this window: P_0 = ip0, P_{i+1} = P_i + s_i, s_{i+1} = step at that time (zstd_fast.c:316-327).
            // pos = the position this lane looks at: P_pair (+1) for the hash lanes, P_{pair+1} for the repcode lane.
            bool const fast = (step == 2) & (sNext == 2) & (ip0 + 2 + ZSA_NP * 2 < nextStep);
            u32 pos, myP = 0, myS = 2, myC = 2, endP, endS, endC, endNext;
            bool valid;
            if (fast) {
                pos = ip0 + fastOff;
                valid = inWin && (ip0 + 2 * pair + 3 < ilimit);            // the pair is entered: `ip3 < ilimit` (:328)
                endP = ip0 + 2 * ZSA_NP; endS = 2; endC = 2; endNext = nextStep;
            } else {
                u32 Pi = ip0, si = sNext, ci = step, ns = nextStep;
#pragma unroll 1
                for (u32 i = 0; i < ZSA_NP; i++) {
                    if (i == pair) { myP = Pi; myS = si; myC = ci; }
                    u32 const Pn = Pi + si, sn = ci;
                    if (Pn + ci >= ns) { ci++; ns += 128; }
                    Pi = Pn; si = sn;
                }
                endP = Pi; endS = si; endC = ci; endNext = ns;
                pos = role == 0 ? myP + myS : myP + (role == 2);
                valid = inWin && (myP + myS + 1 < ilimit);
            }
            // ---- speculative evaluation: every lane compares 4 bytes at `pos` with 4 bytes at `cmpIdx`
            bool const r2lane = chkRep2 && lane == 63;                      // ip0 <= ilimit holds: a window fits
            if (r2lane) { pos = ip0; valid = true; }
            u64 v = 0;
            if (fast) v = havePre ? vPre : zsa_ring64(R, pos - 2);          // every lane: masked ring address
            else if (valid) v = r2lane ? zsa_ring64(R, pos - 2) : RD64(pos);
            havePre = false;
            u32 h = 0, old = 0;
            u32 const myFp = useFp ? ZSA_FPV(v) : 0;
            if (valid && hashLane) { h = zsa_hashT<MLS>(v, hlog); old = T.ld(h); tags[h & (ZSA_NTAGS - 1)] = (u8)lane; }
            // the repcode-type events (role 0, lane 63) compare real bytes; their read does not depend on the table and
            // shares the gather's round trip
            bool repOK = false;
            {   bool const rl = valid && (r2lane || (role == 0 && rep1 > 0));
                u32 const ro = (r2lane ? pos - rep2 : pos - rep1) - 2;
                u32 rv = zsa_ring32(R, ro);                                  // masked ring address, always legal
                if (rl && !(ro >= R.lo && ro + 4 <= R.hi)) rv = wv_ldg32_now(R.src + ro);
                repOK = rl && rv == (u32)v;
            }
            wv_lds_fence();
            u32 const back = (valid && hashLane) ? tags[h & (ZSA_NTAGS - 1)] : (u32)lane;
            if (TAB::filtered && valid && hashLane) tags[h & (ZSA_NTAGS - 1)] = 0;   // (the tags double as the chunk path's bitmap: all zero between uses)
            bool const shared = wv_ballot(back != (u32)lane) != 0;
            u64 const vm = wv_ballot(valid && hashLane);
            u32 cmpIdx = old & posMask, cmpFp = useFp ? old >> posBits : 0;
            if (shared) {   // nearest earlier position of this window in the same cell
                for (u64 m = vm; m; m &= m - 1) {   // ascending: the last hit is the nearest earlier lane
                    int const s = __builtin_ctzll(m);
                    u32 const hs = wv_readlane(h, s), qs = wv_readlane(pos, s), fs = wv_readlane(myFp, s);
                    if (hashLane && s < lane && hs == h) { cmpIdx = qs; cmpFp = fs; }
                }
            }
            bool ok = repOK;
            {   bool const eligible = valid && hashLane && cmpIdx >= prefixStartIndex;
                if (useFp) ok |= eligible && cmpFp == myFp;
                else {   // plain cells: compare the candidate's bytes (ring, or L2 for candidates older than the ring)
                    u32 const co = cmpIdx - 2;
                    u32 cv = zsa_ring32(R, co);
                    if (eligible && !(co >= R.lo && co + 4 <= R.hi)) cv = wv_ldg32_now(R.src + co);
                    ok |= eligible && cv == (u32)v;
                }
            }
            u64 const found = wv_ballot(ok);
            u32 const nv = (u32)__builtin_popcountll(vm) >> 1;
            if (found >> 63) {   // immediate repcode: nothing of the speculative window happened
                u32 const rLength = zsa_count_wave(R, ip0 + 4 - 2, ip0 + 4 - rep2 - 2, iend - 2, lane) + 4;
                { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                {   u32 const v63 = wv_readlane((u32)v, 63);
                    u32 const hh = zsa_hashT<MLS>(v63 | ((u64)wv_readlane((u32)(v >> 32), 63) << 32), hlog);
                    if (lane == 0) T.st(hh, ZSA_CELL(ip0, v63));
                    wv_lds_fence();
                }
                ip0 += rLength;
                if (lane == 0) seqOut[nbSeq] = zsa_packseq(0, rLength - ZSA_MINMATCH, 1);
                ZSA_STAT(5);
                nbSeq++;
                anchor = ip0;
                chkRep2 = (ip0 <= ilimit) && rep2 > 0;
                if (stepSize == 2 && ip0 + 3 < ilimit) {
                    if (R.hi < ip0 - 2 + 512 && R.hi < R.limit) zsa_ring_fill(R, ip0 - 2 + 1024, lane);
                    vPre = zsa_ring64(R, ((chkRep2 && lane == 63) ? ip0 : ip0 + fastOff) - 2); havePre = true;
                }
                    break;   // back to _start with the new ip0
            }
            chkRep2 = false;

            // ---- the event the reference stops at: the first one in lane order that is real
            int w = -1;
            u32 vFwd = 0, vBwd = 0;
            for (u64 cand = found; cand; cand &= cand - 1) {
                int const c = __builtin_ctzll(cand);
                u32 const cPair = (u32)c / 3u, cRole = (u32)c - 3u * cPair;
                if (cRole == 0 || !useFp) { w = c; break; }                 // compared byte for byte already
                u32 const cur = (fast ? ip0 + 2 * cPair : wv_readlane(myP, c)) + (cRole == 2);
                u32 const m = wv_readlane(cmpIdx, c);
                u32 const mb = (cur - anchor) < (m - prefixStartIndex) ? (cur - anchor) : (m - prefixStartIndex);
                zsa_extend_wave(R, cur - 2, m - 2, iend - 2, cur - 2, m - 2, mb, &vFwd, &vBwd, lane);
                if (vFwd >= 4) { w = c; break; }                            // else: fingerprint collision, not an event
            }
            if (w < 0) {
                // all positions of the window enter the table; per cell the last one in window order
                bool dead = false;
                if (shared) {
                    for (u64 m = vm; m; m &= m - 1) {
                        int const s = __builtin_ctzll(m);
                        u32 const hs = wv_readlane(h, s);
                        if (s > lane && hs == h) dead = true;
                    }
                }
                if (valid && hashLane && !dead) T.st(h, ZSA_CELL(pos, v));
                wv_lds_fence();
                if (nv < ZSA_NP || !(endP + endS + 1 < ilimit)) { blockDone = true; break; }
                ip0 = endP; sNext = endS; step = endC; nextStep = endNext;
                continue;
            }
            u32 const wPair = (u32)w / 3u, wRole = (u32)w - 3u * wPair;
            u32 wP, wS, wC;
            if (fast) { wP = ip0 + 2 * wPair; wS = 2; wC = 2; }
            else { wP = wv_readlane(myP, w); wS = wv_readlane(myS, w); wC = wv_readlane(myC, w); }
            u32 const wCand = wv_readlane(cmpIdx, w);
            // table: positions up to and including pair wPair stay (:266, :278, :300-303), plus P_{wPair+1} when the second
            // check hit and step <= 4
            bool const extra = (wRole == 2) && (wC <= 4);
            u32 const xP = wP + wS;                                     // P_{wPair+1}
            bool const xInWin = (wPair + 1 < ZSA_NP) && (wPair + 1 < nv);
            bool const keep = valid && hashLane && (pair <= wPair || (extra && xInWin && pair == wPair + 1 && role == 1));
            {   bool dead = false;
                if (shared) {
                    u64 const km = wv_ballot(keep);
                    for (u64 m = km; m; m &= m - 1) {
                        int const s = __builtin_ctzll(m);
                        u32 const hs = wv_readlane(h, s);
                        if (s > lane && hs == h) dead = true;
                    }
                }
                if (keep && !dead) T.st(h, ZSA_CELL(pos, v));
            }
            wv_lds_fence();
            if (extra && !xInWin) { u64 const vx = RD64(xP); u32 const hx = zsa_hashT<MLS>(vx, hlog); if (lane == 0) T.st(hx, ZSA_CELL(xP, vx)); wv_lds_fence(); }

            u32 current0, match0, mLength, offcode, maxBack;
            if (wRole == 0) {
                current0 = wP;
                ip0 = xP;
                match0 = ip0 - rep1;
                offcode = 1;
                maxBack = 1;                                           // :262 `ip0[-1] == match0[-1]`, unconditional
            } else {
                current0 = wP + (wRole == 2);
                ip0 = current0;
                match0 = wCand;
                rep2 = rep1;
                rep1 = ip0 - match0;
                offcode = rep1 + 3;
                maxBack = (ip0 - anchor) < (match0 - prefixStartIndex) ? (ip0 - anchor) : (match0 - prefixStartIndex);
            }
            if (wRole != 0 && useFp) {   // extended while it was verified
                ip0 -= vBwd; match0 -= vBwd;
                mLength = vBwd + vFwd;
            } else {
                u32 fwd, bwd;
                zsa_extend_wave(R, ip0 + 4 - 2, match0 + 4 - 2, iend - 2, ip0 - 2, match0 - 2, maxBack, &fwd, &bwd, lane);
                ip0 -= bwd; match0 -= bwd;
                mLength = 4 + bwd + fwd;
            }
            if (lane == 0) seqOut[nbSeq] = zsa_packseq(ip0 - anchor, mLength - ZSA_MINMATCH, offcode);
            ZSA_STAT(5);
            nbSeq++;
            ip0 += mLength;
            anchor = ip0;
            if (ip0 <= ilimit) {
                if (R.hi < ip0 - 2 + 512 && R.hi < R.limit) zsa_ring_fill(R, ip0 - 2 + 1024, lane);
                if (stepSize == 2 && ip0 + 3 < ilimit) {   // first window of the next search: step 2, positions ip0 + fastOff
                    vPre = zsa_ring64(R, ((rep2 > 0 && lane == 63) ? ip0 : ip0 + fastOff) - 2); havePre = true;
                }
                {   // :307-309, in this order
                    u32 const pA = current0 + 2, pB = ip0 - 2;
                    u32 const pp = lane == 0 ? pA : pB;
                    u32 hh; u64 wd = 0;
                    if (pA - 2 >= R.lo && ip0 + 4 <= R.hi) { wd = zsa_ring64(R, pp - 2); hh = zsa_hashT<MLS>(wd, hlog); }   // uniform
                    else { hh = 0; if (lane < 2) { wd = RD64(pp); hh = zsa_hashT<MLS>(wd, hlog); } }
                    // pA then pB: when both fall into one cell only the later one is stored
                    bool const sameCell = wv_readlane(hh, 0) == wv_readlane(hh, 1);
                    if (lane < 2 && !(lane == 0 && sameCell)) T.st(hh, ZSA_CELL(pp, wd));
                    wv_lds_fence();
                }
                chkRep2 = rep2 > 0;   // :311-324 immediate repcode: evaluated with the next window
            }
            break;   // goto _start
        }
        if (blockDone) break;
    }
    st.ip0 = ip0; st.anchor = anchor; st.rep1 = rep1; st.rep2 = rep2; st.nbSeq = nbSeq; st.chkRep2 = chkRep2 ? 1u : 0u;
    st.midEpisode = 0; st.nextStep = 0; st.rlo = R.lo; st.rhi = R.hi; st.done = blockDone ? 1u : 0u;
}

// the same, compiled once per (minMatch, table kind) as a function of its own: the chunk path below then owns its registers
template <int MLS, class TAB>
ZSA_DEVFN void zsa_parse_general(const TAB T, u32* ringWords, u8* tags, const ZsaParseCtx* C, ZsaParseState* st, u32 until)
{
    zsa_parse_general_body<MLS, TAB>(T, ringWords, tags, *C, *st, until, wv_lane());
}

// continuation of a match that ran past the first round of zsa_chunk_extend (rare: kept out of line)
ZSA_DEVFN u32 zsa_count_far(const u8* ringB, const u8* src, u32 lo, u32 hi, u32 limit, u32 ip, u32 m, u32 iend)
{
    ZsaRing R; R.b = (u8*)ringB; R.src = src; R.lo = lo; R.hi = hi; R.limit = limit; R.min = 0;
    return zsa_count_wave(R, ip, m, iend, wv_lane());
}
ZSA_DEVFN u32 zsa_back_far(const u8* ringB, const u8* src, u32 lo, u32 hi, u32 limit, u32 ipB, u32 mB, u32 maxBack)
{   // zstd_fast.c:285-289 beyond the first 16 bytes, 64 bytes per step
    ZsaRing R; R.b = (u8*)ringB; R.src = src; R.lo = lo; R.hi = hi; R.limit = limit; R.min = 0;
    int const lane = wv_lane();
    u32 nb = 16;
    while (nb < maxBack) {
        u32 const kk = nb + (u32)lane;
        bool const e = (kk < maxBack) && (zsa_rd8(R, ipB - 1 - kk) == zsa_rd8(R, mB - 1 - kk));
        u64 const ne = wv_ballot(!e);
        u32 const f = ne ? (u32)__builtin_ctzll(ne) : 64u;
        nb += f;
        if (f < 64) break;
    }
    return nb;
}

// ---- CHUNK PATH (levels 1 and 2 without a dictionary; ZsaTabFilt).  While the step is 2 every position from ip0 on is
// visited, so the per-position work that does not depend on the parse state -- source word, hash, fingerprint, probe -- is
// done once for 64 consecutive positions (lane k <-> position winB + k) and reused by every search that starts inside the
// chunk.  The probe reads the LDS filter only; a lane whose filter nibble equals its fingerprint's MAY match.  What remains
// per sequence: a repcode compare (one mask per offset, kept while the offset lives), a scalar pick of the first event in the
// reference's order, ONE global cell read by the winning lane (position + the full fingerprint), the match extension.
// Positions that enter the table are collected in `ins` and written when the chunk is left (commit).
// Exactness: the probe of lane b is what the serial walk sees as long as no earlier lane of the chunk falls into the same
// cell (inserts happen in position order, so only earlier lanes can have written it since set-up).  Set-up finds all lanes
// that share a cell with another one (a bitmap pass over the cell numbers); lanes below the second-lowest of them --
// winValid -- are safe, and the chunk is set up again when the parse gets there.
// Everything the chunk path does not do itself -- the step growing after 128 bytes without a match, the last 288 bytes of a
// block -- is handed back to the caller, which runs the general evaluation up to the next search restart.
// ring fill out of line (about once per KiB of input): returns the new R.hi
ZSA_DEVFN u32 zsa_ring_fill_far(u8* ringB, const u8* src, u32 hi, u32 want, u32 limit)
{
    ZsaRing R; R.b = ringB; R.src = src; R.lo = 0; R.hi = hi; R.limit = limit; R.min = 0;
    if (want > hi + 2 * ZSA_RING_BYTES) R.hi = (want - ZSA_RING_BYTES) & ~1023u;   // a long match: nothing of what it skipped is read again through the ring
    zsa_ring_fill(R, want, wv_lane());
    return R.hi;
}
// The event loop is written so that its loop-carried state stays in scalar registers and every decision is a scalar
// compare and branch: one kind variable instead of flags, masks and offsets updated the same way for every kind of event.
//   M1 / M2: lanes of the chunk whose 4 bytes equal the 4 bytes rep1 / rep2 behind them (0 for an offset of 0).  They are
// made when the chunk is set up; an event then makes the mask of the offset it used -- the read rides on the memory round
// trip of its match extension -- which becomes M1 whatever the kind of event (a repcode event reproduces the mask it used).
template <int MLS>
ZSA_DEVFN void zsa_parse_chunk(const ZsaTabFilt Tin, u32* ringWords, const ZsaParseCtx* Cp, ZsaParseState* stp, u32* backoff)
{
    int const lane = wv_lane();
    ZsaTabFilt T = Tin; T.fb = wv_uniform(Tin.fb);          // (arguments of an out-of-line function count as per-lane values)
    ZsaParseCtx const& C = *Cp; ZsaParseState& st = *stp;
    u32 const hlog = wv_uniform(C.hlog), posBits = wv_uniform(C.posBits), iend = wv_uniform(C.iend), ilimit = iend - 8;
    u32 const prefixStartIndex = wv_uniform(C.prefixStartIndex);
    u64* const seqOut = (u64*)wv_uniform64((u64)C.seqOut);
    bool const useFp = true;
    u32 const posMask = (1u << posBits) - 1u;
    u32 const safeEnd = iend > 288 ? iend - 288 : 0;            // chunks start at or below it: every lane's 8 bytes, all pairs and the
                                                                // first extension round then lie inside the block
    ZsaRing R;
    R.b = (u8*)ringWords; R.src = (const u8*)wv_uniform64((u64)C.base) + 2; R.limit = iend - 2; R.min = 0; R.lo = wv_uniform(st.rlo); R.hi = wv_uniform(st.rhi);
    u32 ip0 = wv_uniform(st.ip0), anchor = wv_uniform(st.anchor), rep1 = wv_uniform(st.rep1), rep2 = wv_uniform(st.rep2), nbSeq = wv_uniform(st.nbSeq);
    u32 chk = wv_uniform(st.chkRep2), mid = wv_uniform(st.midEpisode);
    u32 nextStep = mid ? wv_uniform(st.nextStep) : ip0 + 128;
#ifdef ZSA_PARSE_PROF
    u64 pT = wv_uniform64(((ZsaEncShared*)ringWords)->profLast), pAcc[6] = { 0, 0, 0, 0, 0, 0 };          // rest / set-up / events in the ring / events with an older candidate
#define ZSA_CPROF(i) do { u64 const t_ = wv_clock(); pAcc[i] += t_ - pT; pT = t_; } while (0)
#else
#define ZSA_CPROF(i) do {} while (0)
#endif
    u32* const tabP = T.p;
    u64* seqP = seqOut + nbSeq;                             // where the next sequence goes
    u32 const vz = wv_vzero();                              // a zero the compiler cannot see through: what is or-ed with it is computed by the lanes
    for (;;) {   // one chunk per iteration
        if (nextStep - ip0 <= 6 || ip0 > safeEnd) break;        // the step is about to grow (:322-327), or the block ends
        ZSA_CPROF(0);
        {   u32 const want = ip0 - 2 + 1024;
            if (R.hi < want && R.hi < R.limit) { R.hi = wv_uniform(zsa_ring_fill_far(R.b, R.src, R.hi, want, R.limit)); if (R.hi - R.lo > ZSA_RING_BYTES) R.lo = R.hi - ZSA_RING_BYTES; }
        }
        ZSA_STAT(0);
        ZSA_CPROF(5);
        // ---- set-up: lane k <-> position winB + k.  Memory accesses are issued as early as their addresses are known and
        // consumed late, so that the set-up costs one LDS chain (source word -> filter) and ONE trip to global memory (the
        // cells, and the words behind offsets that reach below the ring), not one trip per item.
        u32 const winB = ip0;
        u32 const q = winB + (u32)lane;
        u64 const wv64 = zsa_ring64(R, q - 2);                  // source word
        // the 4 bytes rep1 / rep2 behind every lane (masks M1 / M2 below): ring addresses are masked, always legal
        u32 rv1 = zsa_ring32(R, q - 2 - rep1), rv2 = zsa_ring32(R, q - 2 - rep2);
        bool const far1 = rep1 != 0 && winB - 2 - rep1 < R.lo, far2 = rep2 != 0 && winB - 2 - rep2 < R.lo;   // uniform: older than the ring
        if (far1) rv1 = wv_ldg32(R.src + (q - 2 - rep1));       // (in the block: chunks start at or below safeEnd, offsets reach no lower than the prefix)
        if (far2) rv2 = wv_ldg32(R.src + (q - 2 - rep2));
        u32 const whash = zsa_hashT<MLS>(wv64, hlog);           // cell number
        u32 const wcell = (ZSA_FPV(wv64) << posBits) | q;       // the lane's own cell value
        u32 const wnib = T.fget(whash);                         // filter bits of the cell at set-up
        u32* const dw = T.dupWord(whash);
        u32 const bit = 1u << (whash & 31u);
        u32 const dprev = wv_atomic_or_lds(dw, bit);            // (shared-cell detection, continued below)
        // The table does not change while the chunk is scanned (commits are deferred), so the cells are fetched here, once per
        // chunk and for all lanes that pass the filter at the same time, instead of once per event in the event's own
        // dependency chain: a cell gives the candidate's position and the fingerprint bits the filter does not hold, and
        // H keeps the lanes whose whole fingerprint agrees.
        bool const pass = wnib == T.top(wcell);                 // the probe passes the filter
        u32 cell = 0;
        if (pass) cell = wv_ldg_cell(tabP + whash);
        // lanes that share their cell with another lane: the first lane to reach a cell's bit finds it clear, the
        // others find it set and clear it again, which tells the first one
        u32 winValid;
        {   bool const late = (dprev & bit) != 0;
            wv_lds_fence();
            (void)wv_atomic_and_lds(dw, late ? ~bit : 0xFFFFFFFFu);
            wv_lds_fence();
            bool const shares = late | ((*dw & bit) == 0);
            wv_lds_fence();
            *dw = 0;
            wv_lds_fence();
            u64 const G = wv_ballot(shares);
            u64 const G2 = G & (G - 1);
            winValid = G2 ? (u32)__builtin_ctzll(G2) : 64u;     // lanes [0, winValid) have exact probes
        }
        if (winValid < 2) { *backoff = ip0 + 1; ZSA_STAT(4); break; }   // (lanes 0 and 1 share a cell: runs of equal bytes)
        u32 const winEnd = winB + (winValid < 63 ? winValid : 63u);
        u32 wcand;                                              // per lane with a table hit: the candidate's position
        u64 H;                                                  // lanes with a table hit
        {   u32 const cx = pass ? cell ^ wcell : 1u << posBits;
            wcand = (cx & posMask) ^ q;                                      // (the low bits of wcell are q)
            H = wv_ballot((cx >> posBits) == 0 && wcand >= prefixStartIndex && (u32)lane < winValid);
        }
        u64 const M1s = wv_ballot(rv1 == (u32)wv64), M2s = wv_ballot(rv2 == (u32)wv64);
        u64 M1 = rep1 ? M1s : 0, M2 = rep2 ? M2s : 0;
        u64 M2c = chk ? M2 : 0;                                 // M2 where the immediate-repcode check is due: after an event (not at a search that continues)
        u32 limEnd = winEnd < nextStep - 3 ? winEnd : nextStep - 3;   // pairs are decided here while both lanes lie below winEnd and the step stays 2
        u32 insV = 0;                                           // per lane: the position entered the table since set-up (not yet written)
        u32 lateA = 0, lateB = 0;                               // positions beyond the chunk that enter the table after what is pending
        ZSA_CPROF(1);
        // ---- the events of the chunk, in the reference's order: the immediate repcode at ip0 (zstd_fast.c:311-324), then per
        // pair the repcode check at ip0 + 2i + 2 and the two table checks
        for (;;) {
            u32 const o = ip0 - winB;
            // kind 0: immediate repcode, 1: repcode at a pair's ip2, 2: table.  cnt: positions from ip0 on that enter the table with
            // the event; pA: the first of the two positions that follow the match into the table (0: none); offBase: as stored
            u32 kind, cur, m, four, eBack, cnt, pA, pBmask, offBase;
            if (((M2c >> o) & 1ull) != 0) { kind = 0; cur = ip0; m = ip0 - rep2; four = 4; eBack = 0; cnt = 1; pA = 0; pBmask = 0; offBase = 1; }
            else {
                u32 const lim = (limEnd - ip0) & ~1u;           // twice the number of pairs decided here, >= 2
                u32 fh = wv_ff1_64(H >> o);                                                            // first table hit (none: 0xFFFFFFFF)
                if (fh >= lim) fh = 64;
                u32 fr = wv_ff1_64((M1 >> o) & 0x5555555555555554ull);                                 // first repcode hit at an ip2
                if (fr > lim) fr = 128;
                if (fr <= fh + 2) { kind = 1; cur = ip0 + fr; m = cur - rep1; four = 4; eBack = 1; cnt = fr; pA = cur; pBmask = 0xFFFFFFFFu; offBase = 1; }
                else if (fh != 64) {
                    kind = 2; cur = ip0 + fh; four = 0; cnt = fh + 2; pA = cur + 2; pBmask = 0xFFFFFFFFu;
                    m = wv_readlane(wcand, (int)(o + fh));      // the candidate position was fetched when the chunk was set up
                    offBase = cur - m + 3;
                    u32 const ba = cur - anchor, bb = m - prefixStartIndex;
                    eBack = ba < bb ? ba : bb;
                } else {   // no event: all pairs enter the table, the search goes on behind the chunk
                    ZSA_STAT(2);
                    wv_cold();
                    insV |= (u32)((u32)lane - o < lim);
                    ip0 += lim;
                    mid = 1; chk = 0;
                    break;
                }
            }
            // one round of match extension: lanes 0..15 compare 8 bytes each forwards from (cur + four, m + four), lanes
            // 48..63 one byte each backwards from (cur - 1, m - 1) for at most eBack bytes (the source side lies in the ring;
            // a candidate older than the ring costs one or two lines).  A repcode's 4 bytes are known to match (:259-265).
            // The same round trip brings the mask of the event's offset.
            u32 const L = cur - winB, off = cur - m;
            u32 fwd, bwd;
            u64 Mn;
            {   bool const fw = lane < 16;
                u32 const k = (u32)lane - 48u;
                u32 const dl = fw ? four + 8u * (u32)lane : 0u - 1u - k;
                u32 const a = cur + dl - 2, b = m + dl - 2;                  // frame offsets
                u64 const sa = zsa_ring64(R, a);                             // masked ring address: legal in every lane
                u32 rv = zsa_ring32(R, q - 2 - off);
                u64 sb;
                if (m - 2 >= R.lo + 16) sb = zsa_ring64(R, b);               // uniform: the candidate lies in the ring as well
                else {   // one trip to global memory for both items (no per-lane branches in this loop: lanes that take no part
                         // read the first needed address)
                    bool const need = fw | ((lane >= 48) & (k < eBack)), needR = (u32)lane >= L;
                    u64 g64 = wv_ldg64(R.src + (need ? b : m + four - 2));
                    u32 g32 = wv_ldg32(R.src + (needR ? q : cur) - 2 - off);
                    wv_join2(g64, g32);
                    sb = need ? g64 : ~sa;
                    rv = needR ? g32 : ~(u32)wv64;
                }
                u64 const diff = sa ^ sb;
                u32 const eq = diff ? ((u32)__builtin_ctzll(diff) >> 3) : 8;
                // a lane stops the count when fewer than `thr` of its bytes agree: 8 forwards; backwards 1 (its byte) within
                // eBack and 9 (always) beyond it.  (Lanes 16..47 are not looked at.)
                u32 const thr = fw ? 8u : (k < eBack ? 1u : 9u);
                u64 const sbal = wv_ballot(eq < thr);
                Mn = wv_ballot(rv == (u32)wv64);
                u32 const sf = (u32)sbal & 0xFFFFu, sk = (u32)(sbal >> 48);
                if (sf) { int const f = __builtin_ctz(sf); fwd = 8u * (u32)f + wv_readlane(eq, f); }
                else fwd = 128 + wv_uniform(zsa_count_far(R.b, R.src, R.lo, R.hi, R.limit, cur + four - 2 + 128, m + four - 2 + 128, iend - 2));   // (a call's result counts as per-lane)
                if (sk) bwd = (u32)__builtin_ctz(sk);
                else bwd = wv_uniform(zsa_back_far(R.b, R.src, R.lo, R.hi, R.limit, cur - 2, m - 2, eBack));
            }
            ZSA_CPROF((m - 2 >= R.lo + 16) ? 2 : 3);
            if (kind == 2 && fwd < 4) { ZSA_STAT(6); H &= ~(1ull << L); continue; }   // the fingerprint let it through, the bytes differ: not an event
            {
                ZSA_STAT(kind == 0 ? 1 : 3);
                u32 const start = cur - bwd, mLength = four + bwd + fwd;
                // (packed by the lanes -- the vector unit has the spare issue slots here -- and stored by every lane, same address,
                // same value: no per-lane branch in this loop)
                wv_stg64(seqP, zsa_packseq((start - anchor) | vz, (mLength - ZSA_MINMATCH) | vz, offBase | vz));   // (an immediate repcode starts at the anchor)
                seqP++;
                // positions that enter the table (:266, :278, :300-303): up to the event's pair, plus the next pair's first position
                // when the second table check hit (step <= 4); an immediate repcode enters its own position (:318).  Then
                // (:307-309) current0 + 2 and ip0 - 2, in this order.
                u32 const ipOld = ip0;
                ip0 = start + mLength;
                u32 pB = (ip0 - 2) & pBmask;
                if (ip0 > ilimit) { pA = 0; pB = 0; }
                insV |= (u32)((q - ipOld < cnt) | (q == pA) | (q == pB));
                // offsets and their masks: the event's offset comes first; a repcode at ip2 keeps the second one
                if (kind != 1) { rep2 = rep1; M2 = M1; }
                rep1 = off; M1 = Mn; M2c = M2;
                anchor = ip0;
                limEnd = winEnd; nextStep = ip0 + 128;          // (the step stays 2 for 128 bytes from here)
                if (ip0 + 2 > winEnd) {                         // no pair of the chunk is left to decide
                    wv_cold();
                    if (pB > winB + 63) { lateB = pB; if (pA > winB + 63) lateA = pA; }   // (only an event that ends the chunk reaches beyond it)
                    mid = 0; chk = (ip0 <= ilimit && rep2 != 0) ? 1u : 0u;
                    break;
                }
            }
        }
        ZSA_CPROF(0);
        // ---- write the positions that entered the table since the chunk was set up.  Lanes below winValid fall into
        // distinct cells that nothing else has written since; the few beyond it follow one by one, in position order
        // (== time order), then the (at most two) positions beyond the chunk
        {   bool const mine = insV != 0;
            if (mine && (u32)lane < winValid) { wv_stg_cell(tabP + whash, wcell); u32 const d = wnib ^ T.top(wcell); if (d) T.fxor(whash, d); }
            for (u64 t = wv_ballot(mine && (u32)lane >= winValid); t; t &= t - 1) {
                wv_lds_fence();
                ZSA_STAT(7);
                if (lane == __builtin_ctzll(t)) { wv_stg_cell(tabP + whash, wcell); u32 const d = T.fget(whash) ^ T.top(wcell); if (d) T.fxor(whash, d); }
            }
            wv_lds_fence();
        }
        if (lateB) {
            if (R.hi < lateB + 16 && R.hi < R.limit) { R.hi = wv_uniform(zsa_ring_fill_far(R.b, R.src, R.hi, lateB + 1024, R.limit)); if (R.hi - R.lo > ZSA_RING_BYTES) R.lo = R.hi - ZSA_RING_BYTES; }
            u32 const pp = lane == 0 ? lateA : lateB;
            u32 hh = 0, cell = 0;
            if (lane < 2) { u64 const wd = zsa_ring64(R, pp - 2); hh = zsa_hashT<MLS>(wd, hlog); cell = ZSA_CELL(pp, wd); }
            if (lane == 0 && lateA) { wv_stg_cell(tabP + hh, cell); u32 const d = T.fget(hh) ^ T.top(cell); if (d) T.fxor(hh, d); }
            wv_lds_fence();
            if (lane == 1) { wv_stg_cell(tabP + hh, cell); u32 const d = T.fget(hh) ^ T.top(cell); if (d) T.fxor(hh, d); }
            wv_lds_fence();
        }
        ZSA_CPROF(4);
    }
#ifdef ZSA_PARSE_PROF
    {   ZsaEncShared* const S_ = (ZsaEncShared*)ringWords;   // (the ring is the first member)
        ZSA_CPROF(0);
        if (lane == 0) { S_->prof[1] += pAcc[0]; S_->prof[8] += pAcc[1]; S_->prof[9] += pAcc[2]; S_->prof[10] += pAcc[3]; S_->prof[7] += pAcc[4]; S_->prof[11] += pAcc[5]; S_->profLast = pT; }
    }
#endif
#undef ZSA_CPROF
    nbSeq = (u32)(seqP - seqOut);
    st.ip0 = ip0; st.anchor = anchor; st.rep1 = rep1; st.rep2 = rep2; st.nbSeq = nbSeq; st.chkRep2 = chk;
    st.midEpisode = mid; st.nextStep = nextStep; st.rlo = R.lo; st.rhi = R.hi; st.done = 0;
}
#undef RD32
#undef RD64
#undef RD8
#undef ZSA_CELL

// Parses one block.  T: the frame's match table in the workgroup's global scratch.  base: window base (index i <-> base + i, the
// frame starts at index 2).  Returns the number of sequences appended to seqOut; rep[] and *lastLL are updated.  All values
// are wave-uniform.  filt / dup: LDS words of the chunk path (null: general evaluation only).
// wholePrefix: a dictionary is loaded right below the block in index space (contiguously, so no extDict segment): the window
// does not limit the prefix (ZSTD_getLowestPrefixIndex with loadedDictEnd != 0) and nothing below the block is read.
template <int MLS>
ZSA_DEV u32 zsa_parse_fast_blockT(u32* T, u32* filt, u32* dup, u32* ringWords, u8* tags, const ZsaEncParams& P, const u8* base, u32 dictLimit,
                                  const u8* src, u32 srcSize, u32 rep[3], u64* seqOut, u32* lastLL, u32 posBits, bool wholePrefix, int lane)
{
    ZsaParseCtx C;
    C.base = base; C.seqOut = seqOut; C.hlog = P.hlog; C.posBits = posBits;
    C.stepSize = (P.tlen > 1) ? (P.tlen + 1) : 2;               // zstd_fast.c:153, :429
    u32 const istart = (u32)(src - base);
    C.iend = istart + srcSize; C.ilimit = C.iend - 8;           // indices
    u32 const maxDistance = wholePrefix ? 0xFFFFFFFFu : 1u << P.wlog;
    // ZSTD_getLowestPrefixIndex (zstd_compress_internal.h:1307-1318)
    C.prefixStartIndex = (C.iend - dictLimit > maxDistance) ? C.iend - maxDistance : dictLimit;
    C.ringMin = wholePrefix ? istart - 2 : 0;
    ZsaParseState st;
    st.anchor = istart; st.ip0 = istart + (istart == C.prefixStartIndex);
    st.rep1 = rep[0]; st.rep2 = rep[1]; st.nbSeq = 0; st.chkRep2 = 0; st.nextStep = 0; st.midEpisode = 0; st.done = 0;
    st.rlo = st.rhi = (istart - 2) & ~1023u;
    u32 saved1 = 0, saved2 = 0;
    {   u32 const curr = st.ip0;
        u32 const windowLow = (curr - dictLimit > maxDistance) ? curr - maxDistance : dictLimit;
        u32 const maxRep = curr - windowLow;
        if (st.rep2 > maxRep) { saved2 = st.rep2; st.rep2 = 0; }
        if (st.rep1 > maxRep) { saved1 = st.rep1; st.rep1 = 0; }
    }
    if (filt) {
        ZsaTabFilt tab; tab.p = T; tab.filt = filt; tab.dup = dup; tab.fb = P.hlog <= 13 ? 4u : 2u;
        u32 backoff = 0;
        for (;;) {
            if (st.ip0 >= backoff) zsa_parse_chunk<MLS>(tab, ringWords, &C, &st, &backoff);
            // whatever stopped the chunk path: the general evaluation takes the parse to the next search restart (or past
            // a stretch where neighbouring positions share cells)
            ZSA_PPROF((ZsaEncShared*)ringWords, lane, 1);
            zsa_parse_general<MLS, ZsaTabFilt>(tab, ringWords, (u8*)dup, &C, &st, backoff > st.ip0 ? backoff : 0u);
            ZSA_PPROF((ZsaEncShared*)ringWords, lane, 0);
            if (wv_uniform(st.done)) break;
        }
    } else {
        ZsaTabGlobal tab; tab.p = T;
        zsa_parse_general<MLS, ZsaTabGlobal>(tab, ringWords, tags, &C, &st, 0xFFFFFFFFu);
    }
    // label _cleanup: zstd_fast.c:330-355
    u32 const rep1 = wv_uniform(st.rep1), rep2 = wv_uniform(st.rep2);
    saved2 = ((saved1 != 0) && (rep1 != 0)) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    *lastLL = C.iend - wv_uniform(st.anchor);
    return wv_uniform(st.nbSeq);
}
#undef ZSA_FPV

ZSA_DEV u32 zsa_parse_fast_block(u32* T, u32* filt, u32* dup, u32* ringWords, u8* tags, const ZsaEncParams& P, const u8* base, u32 dictLimit,
                                 const u8* src, u32 srcSize, u32 rep[3], u64* seqOut, u32* lastLL, u32 posBits, bool wholePrefix, int lane)
{
    switch (P.mml) {
    case 5:  return zsa_parse_fast_blockT<5>(T, filt, dup, ringWords, tags, P, base, dictLimit, src, srcSize, rep, seqOut, lastLL, posBits, wholePrefix, lane);
    case 6:  return zsa_parse_fast_blockT<6>(T, filt, dup, ringWords, tags, P, base, dictLimit, src, srcSize, rep, seqOut, lastLL, posBits, wholePrefix, lane);
    case 7:  return zsa_parse_fast_blockT<7>(T, filt, dup, ringWords, tags, P, base, dictLimit, src, srcSize, rep, seqOut, lastLL, posBits, wholePrefix, lane);
    default: return zsa_parse_fast_blockT<4>(T, filt, dup, ringWords, tags, P, base, dictLimit, src, srcSize, rep, seqOut, lastLL, posBits, wholePrefix, lane);
    }
}
