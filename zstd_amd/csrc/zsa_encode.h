// zsa_encode.h -- device-side Zstandard frame compressor (ZSTD_fast .. ZSTD_lazy2): one 64-lane wave per frame.
//
// Replaces, for independent frames compressed in one launch, what the reference reaches through
// ZSTD_compress / ZSTD_compress_usingCDict (zstd.go:108,115; zstd_bulk.go:77,86).  Output is byte-identical
// to libzstd 1.5.5 for the same cParams (zsa_params.h resolves them, on the host and in the kernels):
//   frame / block loop   zstd_compress.c:4392-4520, 4222-4283, 5107-5144
//   parse                zstd_fast.c:143-403 (noDict), :459-656 (attached dictionary), :658-960 (dictionary as an external
//                        segment); zstd_double_fast.c:98-305; zstd_lazy.c:621-1757
//   literals             zstd_compress_literals.c:40-236, huf_compress.c:147-1406, hist.c:30-165
//   sequences            zstd_compress.c:2637-2986, zstd_compress_sequences.c:157-383, fse_compress.c:69-525
//
// Mapping onto the wave:
//   * one 64-lane wave per frame; the frame's match table lives in the workgroup's slice of the global scratch,
//     LDS holds a source ring, a filter of the table (4 or 2 bits per cell) and a bitmap for shared-cell detection;
//   * every parser evaluates windows of consecutive positions across the lanes and takes the reference's event with one
//     ballot: ZSTD_fast without a dictionary (zsa_parse.h: chunk path + general evaluation), with a dictionary
//     (zsa_parse_dict.h: attached, external segment), ZSTD_dfast (zsa_dfast.h); greedy/lazy/lazy2 evaluate the candidates
//     of one position across lanes (zsa_lazy.h).  No parser runs under `lane == 0`;
//   * literals are gathered by all 64 lanes from the recorded triples (wave prefix sums), histograms use LDS
//     atomics from all lanes, Huffman streams and the FSE sequence bitstream are packed by all lanes;
//   * tables are built by the whole wave: FSE shares and state tables (zsa_common.h: zsa_fse_spread_wave), the Huffman
//     code (ranking into the reference's sort order, code values by ballots).  What stays on lane 0 are dependency
//     chains: the tree's two-queue merge, the quick sort of large-count buckets (its order among equal counts reaches the
//     output), the FSE stream of the Huffman weights, bit-serial table descriptions, headers.
// Serial sections run under `lane == 0` and publish through LDS; every wv_* call is reached by all lanes.
#pragma once
#include "zsa_common.h"
#include "zsa_params.h"

struct ZsaEncParams { u32 wlog, hlog, mml, tlen, clog, strat, slog; };   // strat: 1 ZSTD_fast, 2 ZSTD_dfast, 3 greedy, 4 lazy, 5 lazy2

struct ZsaHufCT { u8 nbBits[256]; u16 val[256]; };
struct ZsaFseCT { u16 nextState[512]; s32 dfs[ZSA_MaxSeq + 1]; u32 dnb[ZSA_MaxSeq + 1]; u32 tableLog; u32 maxSV; };   // maxSV: kept for block-built tables only

struct ZsaCDictDev {            // digested compression dictionary, device resident (zstd_compress.c ZSTD_CDict_s)
    const u8* content; u32 contentSize;   // the part of the dictionary that entered the window (after the entropy header)
    u32 dictSize;               // size of the whole dictionary (cdict->dictContentSize)
    u32 isLoaded;               // 0: dictionary ignored (< 8 bytes)
    u32 dictID;
    s32 level;                  // cdict->compressionLevel
    ZsaCParams cp;              // cpm_createCDict parameters
    const u32* hashTable;       // ZSTD_fast / ZSTD_dfast: tagged cells, (index << 8) | tag, 1 << cp.hlog of them; hash-chain finder: the
                                // chain heads; row finder: the row records (tags, then cells: zsa_lazy.h), 8 << cp.hlog bytes
    const u32* chainTable;      // ZSTD_dfast: the short-hash table (tagged, 1 << cp.clog cells); hash-chain finder: the chain
    u32 useRow;                 // cdict->useRowMatchFinder: greedy .. lazy2 with the CDict's windowLog > 14 (zstd_compress.c:5445-5447)
    u32 rep[3];
    u32 hufRepeat, llRepeat, ofRepeat, mlRepeat;
    ZsaHufCT huf;
    ZsaFseCT ll, of, ml;
};

#ifndef ZSA_RING_BYTES
#define ZSA_RING_BYTES 4096u   // source ring of the parse stage (zsa_parse.h)
#endif
#ifndef ZSA_NTAGS
#define ZSA_NTAGS 1024u       // one-byte tags for shared-cell detection
#endif
struct ZsaHufNode { u32 count; u16 parent; u8 byte; u8 nbBits; };
// scratch of the stream packers
struct ZsaSeqEncScratch { u8 code[3][64]; u16 stBits[3][64]; u8 stNb[3][64]; u32 bitbuf[256]; };

// LDS of one compressing workgroup.  Sized for 16 workgroups per CU (160 KiB / 16): the stages of a block reuse the
// same bytes.  The parse stage takes everything (source ring, the match table's filter, the cell bitmap); the previous
// block's Huffman table -- the one thing of the entropy stages that outlives a block in LDS -- waits in the workgroup's
// global scratch meanwhile.  Below
// ZSTD_lazy the FSE sequence tables of a block are never used again (set_repeat is only reachable with dictionary
// tables, which are re-read from the digest: zstd_compress_sequences.c:180-205); from ZSTD_lazy on a later block may
// price and repeat them (:206-232), so they are parked in the workgroup's global scratch (ZsaEncScratch::park).
#define ZSA_FILT_HLOG_MAX 13u     // the chunk path's LDS filter: 4 bits per cell for match tables of up to 1 << 13 cells,
                                  // 2 bits per cell for 1 << 14 (256 KiB level-1 frames)
#ifndef ZSA_SEQ_HOOK          // test builds (tests/emu) can look at a block's sequences
#define ZSA_SEQ_HOOK(seq, n) do {} while (0)
#endif
struct ZsaEncShared {
    union {
        struct {   // entropy stages
            ZsaHufCT hufPrev, hufNext;
            u32 count[256];                        // histogram of the stage at hand
            union {
                struct {   // Huffman table construction (huf_compress.c:596-766)
                    ZsaHufNode node[2 * 256 + 2];
                    u16 rpBase[192], rpCurr[192];
                };
                struct {   // table descriptions, bit packing, sequence tables
                    union {
                        struct {
                            u8  huffWeight[256];
                            s16 norm[256];
                            u16 cumul[258];
                            u8  tableSymbol[512];
                            // small FSE table for Huffman weights (tableLog <= 6, symbols <= 12)
                            u16 wtNext[64]; s32 wtDfs[16]; u32 wtDnb[16];
                        };
                        ZsaSeqEncScratch enc;      // stream packing (Huffman streams, sequence bitstream)
                    };
                    ZsaFseCT ll, of, ml;
                };
                struct { u32 bLit[64], bSrc[64], bLen[64], bDst[64]; };   // literal gather: unit start, source, length, destination of a run
            };
        };
        struct {   // parse stage (zsa_parse.h): source ring + filter + cell bitmap / shared-cell tags
            u32 ring[ZSA_RING_BYTES / 4 + 4];
            u32 filt[(1u << ZSA_FILT_HLOG_MAX) / 8];
            union { u32 dupBits[(1u << ZSA_FILT_HLOG_MAX) / 32]; u8 tags[ZSA_NTAGS]; };
        };
    };
    // entropy state carried between blocks (zstd_compress_internal.h ZSTD_compressedBlockState_t)
    u32 prevRep[3], nextRep[3];
    u32 prevHufRepeat, nextHufRepeat;
    u32 prevLL, prevOF, prevML, nextLL, nextOF, nextML;   // FSE repeat modes
    u32 err;
    u32 u[24];
    u64 prof[ZSA_NPROF], profLast;
};
static_assert(sizeof(ZsaEncShared) + 8 <= 160 * 1024 / 16, "16 compressing workgroups per CU");
enum { EU_nbSeq, EU_litSize, EU_lastLL, EU_cSize, EU_litCSize, EU_hType, EU_single, EU_hSize, EU_useOld,
       EU_s0, EU_s1, EU_s2, EU_s3, EU_op, EU_tmp, EU_longType, EU_longPos, EU_raw };

// packed sequence record: litLength (18 bits) | mlBase (18 bits) << 18 | offBase (28 bits) << 36
ZSA_DEV u64 zsa_packseq(u32 ll, u32 mlBase, u32 offBase) { return (u64)ll | ((u64)mlBase << 18) | ((u64)offBase << 36); }
ZSA_DEV u32 zsa_seq_ll(u64 s) { return (u32)s & 0x3FFFF; }
ZSA_DEV u32 zsa_seq_ml(u64 s) { return (u32)(s >> 18) & 0x3FFFF; }
ZSA_DEV u32 zsa_seq_of(u64 s) { return (u32)(s >> 36); }

// ------------------------------------------------------------------ LSB-first bit writer (bitstream.h:154-241)
struct ZsaBitW { u8* start; u8* p; u64 acc; u32 nb; };
ZSA_DEV void zsa_bw_init(ZsaBitW& w, u8* dst) { w.start = w.p = dst; w.acc = 0; w.nb = 0; }
ZSA_DEV void zsa_bw_add(ZsaBitW& w, u32 value, u32 nbBits)
{   // nbBits <= 31; BIT_addBits masks the value
    value &= (1u << nbBits) - 1u;
    w.acc |= (u64)value << w.nb;
    w.nb += nbBits;
    while (w.nb >= 8) { *w.p++ = (u8)w.acc; w.acc >>= 8; w.nb -= 8; }
}
ZSA_DEV u32 zsa_bw_close(ZsaBitW& w)
{
    zsa_bw_add(w, 1, 1);
    if (w.nb) { *w.p++ = (u8)w.acc; w.acc = 0; w.nb = 0; }
    return (u32)(w.p - w.start);
}

// ------------------------------------------------------------------ hashing / counting (zstd_compress_internal.h:755-846)
ZSA_DEV u32 zsa_hash(u64 v, u32 hBits, u32 mls)
{   // v = MEM_readLE64(p) (only the low 32 bits matter for mls 4)
    switch (mls) {
    default:
    case 4: return ((u32)v * 2654435761U) >> (32 - hBits);
    case 5: return (u32)(((v << (64 - 40)) * 889523592379ULL) >> (64 - hBits));
    case 6: return (u32)(((v << (64 - 48)) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (u32)(((v << (64 - 56)) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (u32)((v * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}
#include "zsa_parse.h"
#include "zsa_dfast.h"

// window of one frame (zstd_compress_internal.h ZSTD_window_t + the match-state fields the fast parsers read)
struct ZsaWindow {
    const u8* base;       // index i of the current segment <-> base + i
    const u8* dictBase;   // indices in [lowLimit, dictLimit) <-> dictBase + i (extDict)
    u32 dictLimit, lowLimit, loadedDictEnd;
    const ZsaCDictDev* dms;   // attached dictionary, or null
};

#include "zsa_parse_dict.h"   // parse with an attached dictionary (zstd_fast.c:459-656), wave-level
#include "zsa_lazy.h"         // greedy / lazy / lazy2, without a dictionary and with an attached one
#include "zsa_dict_hi.h"      // ZSTD_dfast with an attached dictionary (zstd_double_fast.c:309-532)

// ------------------------------------------------------------------ FSE (fse_compress.c)
ZSA_DEV u32 zsa_fse_minTableLog(u32 srcSize, u32 maxSV)
{
    u32 const a = zsa_highbit32(srcSize) + 1, b = zsa_highbit32(maxSV) + 2;
    return a < b ? a : b;
}
ZSA_DEV u32 zsa_fse_optimalTableLog(u32 maxTableLog, u32 srcSize, u32 maxSV, u32 minus)
{   // fse_compress.c:357-369
    u32 const maxBitsSrc = zsa_highbit32(srcSize - 1) - minus;
    u32 tableLog = maxTableLog;
    u32 const minBits = zsa_fse_minTableLog(srcSize, maxSV);
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < ZSA_FSE_MIN_TABLELOG) tableLog = ZSA_FSE_MIN_TABLELOG;
    if (tableLog > ZSA_FSE_MAX_TABLELOG) tableLog = ZSA_FSE_MAX_TABLELOG;
    return tableLog;
}
// inclusive prefix sum of 64-bit values over the 64 lanes
ZSA_DEV u64 zsa_scan_incl64(u64 v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        u32 const lo = wv_shfl_up((u32)v, d), hi = wv_shfl_up((u32)(v >> 32), d);
        if (lane >= d) v += ((u64)hi << 32) | lo;
    }
    return v;
}
// The second-chance distribution of FSE_normalizeCount (fse_compress.c:379-463: reached when the proportional shares overshoot
// the table by more than half of the largest share), by the whole wave, lane s <-> symbol s.  Symbols whose count is small
// settle for one cell (or the low-probability mark); what the others get is a running total in symbol order -- a 64-bit
// wave prefix sum -- cut at multiples of 2^(62 - tableLog).  Two corner cases end early: every symbol settled (the most
// frequent one takes the cells left over) and nothing left to share out (the cells go round robin to the symbols that own
// cells, in symbol order: a quotient for all and one more for the first few).  count, norm: LDS.  Returns 0 ok, 1 error (uniform).
ZSA_DEV u32 zsa_fse_secondChance_wave(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSV, int lowProbCount, int lane)
{
    bool const mine = (u32)lane <= maxSV;
    u32 const c = mine ? count[lane] : 0u;
    u32 const tableSize = 1u << tableLog;
    u32 const rare = total >> tableLog;                                   // at or below: the low-probability mark
    u32 single = (u32)(((u64)total * 3) >> (tableLog + 1));              // at or below: one cell
    int share = 0;
    bool open = false;                                                    // still waiting for a proportional share
    if (c != 0) { if (c <= rare) share = lowProbCount; else if (c <= single) share = 1; else { share = -2; open = true; } }
    bool const settled = c != 0 && !open;
    u32 given = (u32)__builtin_popcountll(wv_ballot(settled));
    u32 rest = total - wv_reduce_add(settled ? c : 0u);                   // occurrences of the open symbols
    u32 cellsLeft = tableSize - given;
    u32 bad = 0;
    ZSA_STAT(7);
    if (cellsLeft != 0) {
        if ((rest / cellsLeft) > single) {                                // the open symbols are few and heavy: settle the lighter ones too
            single = (u32)(((u64)rest * 3) / (cellsLeft * 2));
            bool const now = open && c <= single;
            if (now) { share = 1; open = false; }
            given += (u32)__builtin_popcountll(wv_ballot(now));
            rest -= wv_reduce_add(now ? c : 0u);
            cellsLeft = tableSize - given;
        }
        if (given == maxSV + 1) {                                         // every symbol settled: the first of the most frequent takes the rest
            u32 const top = wv_reduce_max(c);
            if ((u32)lane == (u32)__builtin_ctzll(wv_ballot(mine && c == top))) share += (int)cellsLeft;
        } else if (rest == 0) {                                           // round robin over the symbols that own cells
            bool const owner = mine && share > 0;
            u64 const om = wv_ballot(owner);
            u32 const owners = (u32)__builtin_popcountll(om), before = (u32)__builtin_popcountll(om & ((1ull << lane) - 1ull));
            if (owner && cellsLeft) share += (int)(cellsLeft / owners + (before < cellsLeft % owners ? 1u : 0u));
        } else {
            u32 const unitLog = 62 - tableLog;
            u64 const half = ((u64)1 << (unitLog - 1)) - 1;
            u64 const perOccurrence = ((((u64)1 << unitLog) * cellsLeft) + half) / rest;
            u64 const mass = open ? (u64)c * perOccurrence : 0;
            u64 const upTo = half + zsa_scan_incl64(mass, lane), from = upTo - mass;
            u32 const cells = (u32)(upTo >> unitLog) - (u32)(from >> unitLog);
            bad = wv_ballot(open && cells < 1) ? 1u : 0u;
            if (open) share = (int)cells;
        }
    }
    if (mine) norm[lane] = (s16)share;
    wv_lds_fence();
    return bad;
}
// FSE_writeNCount (fse_compress.c:234-340) by the whole wave, lane s <-> symbol s.  The description is a concatenation of bit
// fields, one per symbol, whose width and value depend on the cells still unassigned before the symbol -- a prefix sum of the
// cell counts; a symbol without cells that follows another one emits nothing: the first of such a run carries the run's length
// behind its own field (16 one-bits per 24, two one-bits per 3, two bits for the rest).  Every lane makes its field (at most 58
// bits), a prefix sum of the widths places it, the fields are or-ed into 128 bytes of LDS and copied out.  norm: LDS; dst:
// anywhere.  Returns the size in bytes, 0 on error (uniform).  Uses S->cumul.
ZSA_DEV u32 zsa_fse_writeNCount_wave(ZsaEncShared* S, u8* dst, const s16* norm, u32 maxSV, u32 tableLog, int lane)
{
    u32 const tableSize = 1u << tableLog;
    bool const inAlpha = (u32)lane <= maxSV;
    int const n = inAlpha ? (int)norm[lane] : 0;
    u32 const cells = (u32)(n < 0 ? -n : n);
    u32 const upTo = wv_scan_incl(cells), before = upTo - cells;
    if (wv_readlane(upTo, 63) != tableSize) return 0;                     // (:299, :304: the cells must add up to the table)
    u64 const empty = wv_ballot(inAlpha && n == 0);
    bool const inRun = n == 0 && lane > 0 && ((empty >> (lane - 1)) & 1ull) != 0;
    u64 field = 0; u32 width = 0;
    if (inAlpha && before < tableSize && !inRun) {                        // (:274: symbols are described while cells remain)
        u32 const remaining = tableSize + 1 - before, top = zsa_highbit32(remaining), threshold = 1u << top;
        u32 const spare = 2 * threshold - 1 - remaining;                  // values below it take one bit less
        u32 v = (u32)(n + 1);
        if (v >= threshold) v += spare;
        width = top + 1 - (v < spare ? 1u : 0u);
        field = v;
        if (n == 0) {
            u32 const run = (u32)__builtin_ctzll(~(empty >> (lane + 1)));   // empty symbols right behind this one
            u32 const by24 = run / 24, r24 = run - 24 * by24, by3 = r24 / 3, r3 = r24 - 3 * by3;
            field |= (((u64)1 << (16 * by24)) - 1) << width; width += 16 * by24;
            field |= (((u64)1 << (2 * by3)) - 1) << width;   width += 2 * by3;
            field |= (u64)r3 << width;                       width += 2;
        }
    }
    u32 const endBit = wv_scan_incl(width), at = 4 + endBit - width;
    u32 const totalBits = 4 + wv_readlane(endBit, 63);
    u32* const buf = (u32*)S->cumul;
    wv_lds_fence();
    if (lane < 32) buf[lane] = lane == 0 ? tableLog - ZSA_FSE_MIN_TABLELOG : 0u;
    wv_lds_fence();
    if (width) {
        u32 const w = at >> 5, sh = at & 31u;
        u64 const lo = field << sh;
        u32 const hi = sh ? (u32)(field >> (64 - sh)) : 0u;
        (void)wv_atomic_or_lds(&buf[w], (u32)lo);
        if ((u32)(lo >> 32)) (void)wv_atomic_or_lds(&buf[w + 1], (u32)(lo >> 32));
        if (hi) (void)wv_atomic_or_lds(&buf[w + 2], hi);
    }
    wv_lds_fence();
    u32 const size = (totalBits + 7) >> 3;
    for (u32 i = (u32)lane; i < size; i += ZSA_WAVE) dst[i] = ((const u8*)buf)[i];
    wv_lds_fence();
    return size;
}
// The same normalisation by the whole wave, lane s <-> symbol s (sequence codes: at most 53 symbols).  count, norm: LDS.
// Every symbol's share is independent of the others (fse_compress.c:487-508); what couples them -- the cells left over, the
// first symbol with the largest share, which absorbs them -- is a wave sum, a wave maximum and a ballot.  The rare
// second-chance distribution is zsa_fse_secondChance_wave above.  Returns 0 ok, 1 error; wave-uniform.
ZSA_DEV u32 zsa_fse_normalizeCount_wave(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSV, u32 useLowProb, int lane)
{
    if (tableLog < ZSA_FSE_MIN_TABLELOG || tableLog > ZSA_FSE_MAX_TABLELOG) return 1;
    if (tableLog < zsa_fse_minTableLog(total, maxSV)) return 1;
    bool const mine = (u32)lane <= maxSV;
    u32 const c = mine ? count[lane] : 0;
    if (wv_ballot(mine && c == total)) return 0;                 // one symbol only: the caller has chosen RLE (:488)
    int const lowProbCount = useLowProb ? -1 : 1;
    u64 const scale = 62 - tableLog, step = ((u64)1 << 62) / total, vStep = (u64)1 << (scale - 20);
    u32 const lowThreshold = total >> tableLog;
    int share = 0; u32 cells = 0; bool regular = false;
    if (c != 0) {
        if (c <= lowThreshold) { share = lowProbCount; cells = 1; }
        else {
            u32 p = (u32)(((u64)c * step) >> scale);
            if (p < 8) {   // round up when the remainder beats the threshold of this small share (:494-497)
                u32 const rtb = p == 0 ? 0u : (p == 1 ? 473195u : (p == 2 ? 504333u : (p == 3 ? 520860u : (p == 4 ? 550000u : (p == 5 ? 700000u : (p == 6 ? 750000u : 830000u))))));
                p += ((u64)c * step) - ((u64)p << scale) > vStep * rtb;
            }
            share = (int)(s16)p; cells = p; regular = true;
        }
    }
    if (mine) norm[lane] = (s16)share;
    int const still = (int)(1u << tableLog) - (int)wv_reduce_add(cells);
    u32 const top = wv_reduce_max(regular && share > 0 ? (u32)share : 0u);
    u64 const tm = wv_ballot(regular && share > 0 && (u32)share == top);
    u32 const largest = tm ? (u32)__builtin_ctzll(tm) : 0u;
    wv_lds_fence();
    int const atLargest = norm[largest];
    wv_lds_fence();
    if (-still >= (atLargest >> 1)) return zsa_fse_secondChance_wave(norm, tableLog, count, total, maxSV, lowProbCount, lane);
    if ((u32)lane == largest) norm[largest] = (s16)(atLargest + still);
    wv_lds_fence();
    return 0;
}
// FSE_buildCTable_wksp (fse_compress.c:69-215) by the whole wave: symbol s's transform and first state from a wave prefix sum
// of the cell counts, the state table from zsa_fse_spread_wave (zsa_common.h).  norm: LDS; nextState/dfs/dnb: LDS or global.
ZSA_DEV void zsa_fse_buildCTable_wave(ZsaEncShared* S, u16* nextState, s32* dfs, u32* dnb, const s16* norm, u32 maxSV, u32 tableLog, int lane)
{
    u32 const tableSize = 1u << tableLog;
    int const nc = (u32)lane <= maxSV ? (int)norm[lane] : 0;
    u32 const cells = nc == -1 ? 1u : (u32)nc;
    u32 const before = wv_scan_incl(cells) - cells;              // cells of the symbols below this one
    u16* const firstState = S->cumul + 128;
    firstState[lane] = (u16)before;
    if ((u32)lane <= maxSV) {   // symbol transform (:178-201)
        if (nc == 0) { dnb[lane] = ((tableLog + 1) << 16) - tableSize; dfs[lane] = 0; }
        else if (nc == -1 || nc == 1) { dnb[lane] = (tableLog << 16) - tableSize; dfs[lane] = (s32)(before - 1); }
        else {
            u32 const maxBitsOut = tableLog - zsa_highbit32((u32)nc - 1);
            dnb[lane] = (maxBitsOut << 16) - ((u32)nc << maxBitsOut);
            dfs[lane] = (s32)(before - (u32)nc);
        }
    }
    ZsaSpreadScratch X; X.first = S->cumul; X.seen = S->cumul + 64; X.sym = S->tableSymbol;
    zsa_fse_spread_wave(X, norm, maxSV, tableLog, lane,
                        [&](u32 cell, u32 sy, u32 rank) { nextState[firstState[sy] + rank] = (u16)(tableSize + cell); });
    wv_lds_fence();
}
// FSE encoder state (fse.h:437-476)
struct ZsaFseCState { u32 value; const u16* next; const s32* dfs; const u32* dnb; u32 stateLog; };
ZSA_DEV void zsa_fse_initCState2(ZsaFseCState& st, const u16* next, const s32* dfs, const u32* dnb, u32 tableLog, u32 symbol)
{
    u32 const nbBitsOut = (dnb[symbol] + (1u << 15)) >> 16;
    st.next = next; st.dfs = dfs; st.dnb = dnb; st.stateLog = tableLog;
    st.value = (nbBitsOut << 16) - dnb[symbol];
    st.value = next[(s32)(st.value >> nbBitsOut) + dfs[symbol]];
}
ZSA_DEV void zsa_fse_encodeSymbol(ZsaBitW& w, ZsaFseCState& st, u32 symbol)
{
    u32 const nbBitsOut = (st.value + st.dnb[symbol]) >> 16;
    zsa_bw_add(w, st.value, nbBitsOut);
    st.value = st.next[(s32)(st.value >> nbBitsOut) + st.dfs[symbol]];
}

// ------------------------------------------------------------------ Huffman table construction (huf_compress.c), serial
ZSA_DEV u32 zsa_huf_bucket(u32 count) { return (count < 166) ? count : zsa_highbit32(count) + 158; }
// Orders ranks [first, last] of one large-count bucket by descending count the way the reference's sort does
// (huf_compress.c:531-583).  That sort is not stable, and which of two equally frequent symbols ends up first decides which
// gets the shorter code, so the result has to be reproduced element move for element move: spans of fewer than 8 elements are
// built up by insertion; longer ones are split around the count of their last element (larger counts swapped to the front in
// the order they are met), the shorter part becomes a span of its own and is finished first, the longer part is split on
// without being tested for "short" again.  Device code has no recursion: the spans wait on a small stack (depth <= log2(256)).
ZSA_DEV void zsa_huf_orderBucket(ZsaHufNode* a, int first, int last)
{
    int lo[12], hi[12]; bool untested[12];
    int depth = 1;
    lo[0] = first; hi[0] = last; untested[0] = true;
    while (depth > 0) {
        int const t = depth - 1;
        if (untested[t]) {
            untested[t] = false;
            if (hi[t] - lo[t] < 8) {
                for (int i = lo[t] + 1; i <= hi[t]; i++) {
                    ZsaHufNode const x = a[i];
                    int j = i;
                    while (j > lo[t] && a[j - 1].count < x.count) { a[j] = a[j - 1]; j--; }
                    a[j] = x;
                }
                depth--;
                continue;
            }
        }
        if (lo[t] >= hi[t]) { depth--; continue; }
        u32 const pivot = a[hi[t]].count;
        int w = lo[t];                                           // where the next larger count goes
        for (int r = lo[t]; r < hi[t]; r++)
            if (a[r].count > pivot) { ZsaHufNode const x = a[w]; a[w] = a[r]; a[r] = x; w++; }
        { ZsaHufNode const x = a[w]; a[w] = a[hi[t]]; a[hi[t]] = x; }
        if (w - lo[t] < hi[t] - w) { lo[depth] = lo[t]; hi[depth] = w - 1; lo[t] = w + 1; }
        else { lo[depth] = w + 1; hi[depth] = hi[t]; hi[t] = w - 1; }
        untested[depth] = true;
        depth++;
    }
}
// HUF_sort (huf_compress.c:596-641) by the whole wave.  The reference drops the symbols into buckets (counts below 166: one
// bucket per count; above: one per power of two), buckets in descending order and symbols in ascending order inside a bucket,
// and then sorts the buckets of large counts with its quick sort.  Here lane l owns symbols l, l + 64, l + 128, l + 192; a
// symbol's slot is the number of symbols whose key (bucket, 255 - symbol) is larger, counted against the keys of all lanes
// (readlane, no memory), and lane 0 then runs the quick sort over each large-count bucket with more than one member -- its
// order among equal counts is part of the format's byte identity, so that step is kept move for move.
// Returns the rank of the last symbol with a non-zero count.
ZSA_DEV u32 zsa_huf_sort_wave(ZsaHufNode* node, const u32* count, u32 maxSV, int lane)
{
    u32 const maxSV1 = maxSV + 1;
    u32 key[4], cnt[4], pos[4];
    for (u32 k = 0; k < 4; k++) {
        u32 const n = (u32)lane + 64u * k;
        cnt[k] = n < maxSV1 ? count[n] : 0u;
        key[k] = n < maxSV1 ? ((zsa_huf_bucket(cnt[k]) + 1u) << 8) | (255u - n) : 0u;   // (0: not a symbol, sorts last)
        pos[k] = 0;
    }
    for (u32 g = 0; g < 4 && 64u * g < maxSV1; g++) {
        u32 const lim = maxSV1 - 64u * g < 64u ? maxSV1 - 64u * g : 64u;
        for (u32 j = 0; j < lim; j++) {
            u32 const other = wv_readlane(key[g], (int)j);
            for (u32 k = 0; k < 4; k++) pos[k] += other > key[k] ? 1u : 0u;
        }
    }
    for (u32 k = 0; k < 4; k++) if (key[k]) { node[pos[k]].count = cnt[k]; node[pos[k]].byte = (u8)((u32)lane + 64u * k); }
    u32 nonNull = 0;
    for (u32 k = 0; k < 4; k++) nonNull += (u32)__builtin_popcountll(wv_ballot(cnt[k] != 0));
    wv_sync();
    // the buckets of large counts, largest first: [start, start + size)
    // (bucket 165 holds the counts 165 .. 255, 166 the counts 256 .. 511, ...: huf_compress.c:630-638 sorts from there up)
    u32 start = 0;
    for (u32 b = 190; b >= 165; b--) {
        u32 size = 0;
        for (u32 k = 0; k < 4; k++) size += (u32)__builtin_popcountll(wv_ballot((key[k] >> 8) == b + 1u));
        if (size > 1 && lane == 0) zsa_huf_orderBucket(node + start, 0, (int)size - 1);
        start += size;
    }
    wv_sync();
    return nonNull - 1;
}
// HUF_buildTree (huf_compress.c:657-694): the two-queue merge is one dependency chain (lane 0); the sentinels before it and the
// leaves' depths after it are set by all lanes.
ZSA_DEV void zsa_huf_buildTree_wave(ZsaHufNode* node, u32 nonNullRank, int lane)
{
    ZsaHufNode* const node0 = node - 1;
    int const nodeRoot = 256 + (int)nonNullRank - 1;
    for (int n = 257 + lane; n <= nodeRoot; n += ZSA_WAVE) node[n].count = 1u << 30;
    if (lane == 0) node0[0].count = 1u << 31;
    wv_sync();
    if (lane == 0) {
        int lowS = (int)nonNullRank, lowN = 256, nodeNb = 256;
        node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
        node[lowS].parent = node[lowS - 1].parent = (u16)nodeNb;
        nodeNb++; lowS -= 2;
        while (nodeNb <= nodeRoot) {
            int const n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
            int const n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
            node[nodeNb].count = node[n1].count + node[n2].count;
            node[n1].parent = node[n2].parent = (u16)nodeNb;
            nodeNb++;
        }
        node[nodeRoot].nbBits = 0;
        for (int n = nodeRoot - 1; n >= 256; n--) node[n].nbBits = node[node[n].parent].nbBits + 1;
    }
    wv_sync();
    for (int n = lane; n <= (int)nonNullRank; n += ZSA_WAVE) node[n].nbBits = node[node[n].parent].nbBits + 1;
    wv_sync();
}
// Depth limit of the Huffman tree (huf_compress.c:352-474), nodes in rank order (descending count, depths non-decreasing).
// By all lanes: every leaf deeper than the limit is cut to it, which over-subscribes the code space by a debt counted in
// units of the deepest code -- a wave sum; the last rank of every depth below the limit -- where a leaf can be pushed one
// level down to repay 2^(steps below the limit - 1) units -- is a ballot per depth.  The repayment itself is a short chain of
// decisions that each read what the previous one changed (a handful of steps): wave-uniform scalar work, stores by lane 0.
// lastAt: LDS, ZSA_HUF_TABLELOG_MAX + 2 entries.  Returns the depth of the deepest leaf afterwards (uniform).
ZSA_DEV u32 zsa_huf_limitDepth_wave(ZsaHufNode* node, u16* lastAt, u32 lastRank, u32 limit, int lane)
{
    u32 const deepest = node[lastRank].nbBits;
    if (deepest <= limit) return deepest;
    u32 const NONE = 0xFFFFu;
    // cut, and count the debt
    u32 depth[4], owed = 0;
    wv_lds_fence();
    for (u32 k = 0; k < 4; k++) {
        u32 const r = (u32)lane + 64u * k;
        depth[k] = r <= lastRank ? node[r].nbBits : 0xFFu;
        if (depth[k] != 0xFFu && depth[k] > limit) { owed += (1u << (deepest - limit)) - (1u << (deepest - depth[k])); node[r].nbBits = (u8)limit; depth[k] = limit; }
    }
    int debt = (int)(wv_reduce_add(owed) >> (deepest - limit));
    // the last rank of every depth below the limit, and the last rank above which everything sits at the limit
    u32 below = 0;                                               // highest rank with a depth below the limit
    for (u32 d = 1; d <= ZSA_HUF_TABLELOG_MAX + 1; d++) {
        u32 at = NONE;
        if (d < limit) for (int k = 3; k >= 0; k--) {
            u64 const m = wv_ballot(depth[k] == limit - d);
            if (m) { at = 64u * (u32)k + 63u - (u32)__builtin_clzll(m); break; }
        }
        if (at != NONE && at > below) below = at;
        if ((u32)lane == d) lastAt[d] = (u16)at;
    }
    if (lane == 0) lastAt[0] = (u16)NONE;
    wv_lds_fence();
    // repay: lengthen the cheapest suitable leaf -- starting from the largest step that does not overshoot, a smaller step is
    // preferred while the leaf it would push down is less than twice as frequent as the one of the larger step
    while (debt > 0) {
        u32 d = zsa_highbit32((u32)debt) + 1;
        for (; d > 1; d--) {
            u32 const big = lastAt[d], small = lastAt[d - 1];
            if (big == NONE) continue;
            if (small == NONE) break;
            if (node[big].count <= 2 * node[small].count) break;
        }
        while (d <= ZSA_HUF_TABLELOG_MAX && lastAt[d] == NONE) d++;
        u32 const r = lastAt[d];
        debt -= 1 << (d - 1);
        u32 const under = r != 0 ? (u32)node[r - 1].nbBits : 0xFFu;   // (read before the store below: ranks differ)
        wv_lds_fence();
        if (lane == 0) {
            node[r].nbBits++;                                    // rank r now sits d - 1 steps below the limit -- and is the last one there
            if (lastAt[d - 1] == NONE) lastAt[d - 1] = (u16)r;
            lastAt[d] = (u16)((r != 0 && under == limit - d) ? r - 1 : NONE);
        }
        wv_lds_fence();
    }
    // overpaid (the steps are powers of two): take leaves at the limit one level up again, highest ranks first
    while (debt < 0) {
        u32 r;
        if (lastAt[1] == NONE) { while (node[below].nbBits == limit) below--; r = below + 1; }
        else r = (u32)lastAt[1] + 1;
        wv_lds_fence();
        if (lane == 0) { node[r].nbBits--; lastAt[1] = (u16)r; }
        wv_lds_fence();
        debt++;
    }
    return limit;
}
// HUF_buildCTable_wksp (huf_compress.c:730-766 + :706-728) by all lanes; returns maxNbBits (0 on error), uniform.
// Code values: symbols of one length take consecutive values in symbol order -- the rank of a symbol among the symbols of its
// length is a ballot per length over 64 symbols at a time.
ZSA_DEV u32 zsa_huf_buildCTable_wave(ZsaEncShared* S, ZsaHufCT* ct, const u32* count, u32 maxSV, u32 maxNbBits, int lane)
{
    ZsaHufNode* const node = S->node + 1;
    for (u32 i = (u32)lane; i < 2 * 256 + 2; i += ZSA_WAVE) { S->node[i].count = 0; S->node[i].parent = 0; S->node[i].byte = 0; S->node[i].nbBits = 0; }
    wv_sync();
    u32 const nonNullRank = zsa_huf_sort_wave(node, count, maxSV, lane);
    zsa_huf_buildTree_wave(node, nonNullRank, lane);
    maxNbBits = zsa_huf_limitDepth_wave(node, S->rpBase, nonNullRank, maxNbBits, lane);
    wv_sync();
    if (maxNbBits > ZSA_HUF_TABLELOG_MAX) return 0;
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { ct->nbBits[i] = 0; ct->val[i] = 0; }
    wv_sync();
    for (u32 n = (u32)lane; n <= maxSV; n += ZSA_WAVE) ct->nbBits[node[n].byte] = node[n].nbBits;
    wv_sync();
    // first value per length (:712-722): from the number of symbols of each length among ranks 0 .. nonNullRank
    u32 nb[4];
    for (u32 k = 0; k < 4; k++) { u32 const n = (u32)lane + 64u * k; nb[k] = n <= maxSV ? ct->nbBits[n] : 0xFFu; }
    u32 valPerRank[ZSA_HUF_TABLELOG_MAX + 1];
    {   u32 perRank[ZSA_HUF_TABLELOG_MAX + 1];
        for (u32 b = 0; b <= ZSA_HUF_TABLELOG_MAX; b++) {
            u32 c = 0;
            if (b) for (u32 k = 0; k < 4; k++) c += (u32)__builtin_popcountll(wv_ballot(nb[k] == b));   // (symbols of length b all have a count: they lie below nonNullRank)
            perRank[b] = c;
        }
        u32 min = 0;   // (the reference's U16 arithmetic)
        valPerRank[0] = 0;
        for (u32 n = ZSA_HUF_TABLELOG_MAX; n > 0; n--) {
            valPerRank[n] = 0;
            if (n <= maxNbBits) { valPerRank[n] = min & 0xFFFFu; min = ((min + perRank[n]) & 0xFFFFu) >> 1; }
        }
    }
    for (u32 k = 0; k < 4; k++) {
        u32 v = 0;
        for (u32 b = 1; b <= ZSA_HUF_TABLELOG_MAX; b++) {   // (no symbol is longer than maxNbBits)
            u64 const m = wv_ballot(nb[k] == b);
            if (nb[k] == b) v = valPerRank[b] + (u32)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            valPerRank[b] += (u32)__builtin_popcountll(m);
        }
        u32 const n = (u32)lane + 64u * k;
        if (n <= maxSV) ct->val[n] = (u16)v;
    }
    wv_sync();
    return maxNbBits;
}
// HUF_compressWeights (huf_compress.c:147-187) by the whole wave: histogram of the weights by ballots (13 values), shares and
// encode table by the wave routines the sequence section uses; the description's bits and the stream itself -- two interleaved
// states, one chain -- on lane 0.  weights: LDS.  Returns size; 0 "not compressible", 1 "rle", 0xFFFFFFFF error.  Uniform.
ZSA_DEV u32 zsa_huf_compressWeights_wave(ZsaEncShared* S, u8* dst, const u8* weights, u32 wtSize, int lane)
{
    if (wtSize <= 1) return 0;
    u32 w[4];
    for (u32 k = 0; k < 4; k++) { u32 const n = (u32)lane + 64u * k; w[k] = n < wtSize ? weights[n] : 0xFFu; }
    u32 myCount = 0;   // lane t: how many symbols have weight t
    for (u32 t = 0; t <= ZSA_HUF_TABLELOG_MAX; t++) {
        u32 c = 0;
        for (u32 k = 0; k < 4; k++) c += (u32)__builtin_popcountll(wv_ballot(w[k] == t));
        if ((u32)lane == t) myCount = c;
    }
    u32 const maxSV = 63u - (u32)__builtin_clzll(wv_ballot(myCount != 0));
    u32 const maxCount = wv_reduce_max(myCount);
    if (maxCount == wtSize) return 1;
    if (maxCount == 1) return 0;
    u32 const tableLog = zsa_fse_optimalTableLog(6, wtSize, maxSV, 2);
    u32* const wcount = (u32*)&S->norm[64];   // (16 words behind the shares)
    if (lane < 16) wcount[lane] = myCount;
    wv_lds_fence();
    if (zsa_fse_normalizeCount_wave(S->norm, tableLog, wcount, wtSize, maxSV, 0, lane)) return 0xFFFFFFFFu;
    wv_lds_fence();
    u32 const h = zsa_fse_writeNCount_wave(S, dst, S->norm, maxSV, tableLog, lane);
    if (h == 0) return 0xFFFFFFFFu;
    zsa_fse_buildCTable_wave(S, S->wtNext, S->wtDfs, S->wtDnb, S->norm, maxSV, tableLog, lane);
    wv_sync();
    if (wtSize <= 2) return 0;
    u32 n = 0;
    if (lane == 0) {   // fse_compress.c:551-608 FSE_compress_usingCTable_generic
        u8* const op = dst + h;
        const u8* ip = weights + wtSize;
        u32 srcSize = wtSize;
        ZsaBitW bw;
        ZsaFseCState s1, s2;
        zsa_bw_init(bw, op);
        if (srcSize & 1) {
            zsa_fse_initCState2(s1, S->wtNext, S->wtDfs, S->wtDnb, tableLog, *--ip);
            zsa_fse_initCState2(s2, S->wtNext, S->wtDfs, S->wtDnb, tableLog, *--ip);
            zsa_fse_encodeSymbol(bw, s1, *--ip);
        } else {
            zsa_fse_initCState2(s2, S->wtNext, S->wtDfs, S->wtDnb, tableLog, *--ip);
            zsa_fse_initCState2(s1, S->wtNext, S->wtDfs, S->wtDnb, tableLog, *--ip);
        }
        srcSize -= 2;
        if (srcSize & 2) { zsa_fse_encodeSymbol(bw, s2, *--ip); zsa_fse_encodeSymbol(bw, s1, *--ip); }
        while (ip > weights) {
            zsa_fse_encodeSymbol(bw, s2, *--ip); zsa_fse_encodeSymbol(bw, s1, *--ip);
            zsa_fse_encodeSymbol(bw, s2, *--ip); zsa_fse_encodeSymbol(bw, s1, *--ip);
        }
        zsa_bw_add(bw, s2.value, s2.stateLog);
        zsa_bw_add(bw, s1.value, s1.stateLog);
        n = h + zsa_bw_close(bw);
    }
    return wv_readlane(n, 0);
}
// huf_compress.c:230-268 HUF_writeCTable_wksp; returns header size, 0 on error (uniform).  All lanes.
ZSA_DEV u32 zsa_huf_writeCTable(ZsaEncShared* S, u8* dst, const ZsaHufCT* ct, u32 maxSV, u32 huffLog, int lane)
{
    u8* const huffWeight = S->huffWeight;
    for (u32 n = (u32)lane; n < maxSV; n += ZSA_WAVE) { u32 const b = ct->nbBits[n]; huffWeight[n] = (u8)(b ? huffLog + 1 - b : 0); }
    if (lane == 0) huffWeight[maxSV] = 0;
    wv_sync();
    u32 const hSize = zsa_huf_compressWeights_wave(S, dst + 1, huffWeight, maxSV, lane);
    wv_sync();
    if (hSize == 0xFFFFFFFFu) return 0;
    if ((hSize > 1) & (hSize < maxSV / 2)) { if (lane == 0) dst[0] = (u8)hSize; return hSize + 1; }
    if (maxSV > (256 - 128)) return 0;
    if (lane == 0) dst[0] = (u8)(128 + (maxSV - 1));
    for (u32 n = 2u * (u32)lane; n < maxSV; n += 2u * ZSA_WAVE) dst[(n / 2) + 1] = (u8)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSV + 1) / 2) + 1;
}
ZSA_DEV u32 zsa_huf_estimate(const ZsaHufCT* ct, const u32* count, u32 maxSV, int lane)
{   // huf_compress.c:768-777 (a wave sum; a block's literals are at most 128 KiB, so the bit count fits 32 bits)
    u32 nbBits = 0;
    for (u32 s = (u32)lane; s <= maxSV; s += ZSA_WAVE) nbBits += (u32)ct->nbBits[s] * count[s];
    return wv_reduce_add(nbBits) >> 3;
}
// The same stream written by all lanes: 256 symbols per step (4 per lane, last symbol first), prefix sum of the code
// lengths, LDS atomic-or packing, whole words flushed together.  Returns the stream size (uniform).
#define ZSA_HUF_NOFIT 0xFFFFFFFFu
ZSA_DEV u32 zsa_huf_encode_wave(ZsaEncShared* S, u8* dst, const u8* src, u32 srcSize, const ZsaHufCT* ct, u32 cap, int lane);

// all-lane histogram of bytes into S->count[0..255] (hist.c:30-165): exact counts
ZSA_DEV void zsa_hist_bytes(ZsaEncShared* S, const u8* src, u32 n, int lane)
{
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) S->count[i] = 0;
    wv_sync();
    for (u32 i = (u32)lane; i < n; i += ZSA_WAVE) wv_atomic_add_lds(&S->count[src[i]], 1);
    wv_sync();
}

// One Huffman stream (huf_compress.c:1023-1086), at most `cap` bytes: a stream that would be longer is given up as soon as that
// is known and ZSA_HUF_NOFIT returned -- the reference's writer stops at the end of its buffer the same way (:885-950) and the
// literals then go out raw; with a table taken over from a dictionary a stream can be a third LONGER than its input, which
// nothing else keeps inside the frame's destination.
ZSA_DEV u32 zsa_huf_encode_wave(ZsaEncShared* S, u8* dst, const u8* src, u32 srcSize, const ZsaHufCT* ct, u32 cap, int lane)
{
    ZsaSeqEncScratch* const E = &S->enc;
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) E->bitbuf[i] = 0;
    u32 totalBits = 0, flushedWords = 0;
    // (the table is one of two in LDS, chosen at run time: as LDS pointers; the literals lie in global scratch: a lane's four
    // symbols are one 4-byte load)
    ZSA_LDS const u8* const nbT = (ZSA_LDS const u8*)ct->nbBits;
    ZSA_LDS const u16* const valT = (ZSA_LDS const u16*)ct->val;
    wv_lds_fence();
    for (u32 tBase = 0; tBase < srcSize; tBase += 4 * ZSA_WAVE) {
        u64 blob = 0; u32 len = 0;
        u32 const t0 = tBase + 4u * (u32)lane;
        if (t0 + 4 <= srcSize) {
            u32 const w4 = wv_ldg32(src + (srcSize - 4 - t0));   // symbols t0 .. t0 + 3 are its bytes 3 .. 0
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
                u32 const sym = (w4 >> (8u * (3u - k))) & 0xFFu;
                blob |= (u64)valT[sym] << len;
                len += nbT[sym];
            }
        } else
        for (u32 k = 0; k < 4; k++) {
            u32 const t = t0 + k;
            if (t < srcSize) {
                u32 const sym = wv_ldg8(src + (srcSize - 1 - t));
                blob |= (u64)valT[sym] << len;
                len += nbT[sym];
            }
        }
        u32 const incl = wv_scan_incl(len);
        u32 const chunkBits = wv_readlane(incl, 63);
        if (len) {
            u32 const rel = totalBits + (incl - len) - flushedWords * 32;
            u32 const w = rel >> 5, sh = rel & 31;
            u64 const nlo = blob << sh;
            u32 const nhi = sh ? (u32)(blob >> (64 - sh)) : 0;
            if ((u32)nlo) wv_atomic_or_lds(&E->bitbuf[w], (u32)nlo);
            if ((u32)(nlo >> 32)) wv_atomic_or_lds(&E->bitbuf[w + 1], (u32)(nlo >> 32));
            if (nhi) wv_atomic_or_lds(&E->bitbuf[w + 2], nhi);
        }
        totalBits += chunkBits;
        wv_lds_fence();
        if (((totalBits + 8) >> 3) > cap) return ZSA_HUF_NOFIT;   // (with the end mark; uniform; whole words only are flushed below: never past cap)
        {   u32 const done = (totalBits >> 5) - flushedWords;
            for (u32 i = (u32)lane; i < done; i += ZSA_WAVE) wv_stg32(dst + 4 * (flushedWords + i), E->bitbuf[i]);
            u32 const carry = E->bitbuf[done];
            wv_lds_fence();
            for (u32 i = (u32)lane; i <= done + 4 && i < 256; i += ZSA_WAVE) E->bitbuf[i] = 0;
            wv_lds_fence();
            if (lane == 0) E->bitbuf[0] = carry;
            flushedWords += done;
            wv_lds_fence();
        }
    }
    // HUF_closeCStream (huf_compress.c:941-950): end mark, flush
    if (lane == 0) { u32 const rel = totalBits - flushedWords * 32; E->bitbuf[rel >> 5] |= 1u << (rel & 31); }
    totalBits += 1;
    wv_lds_fence();
    u32 const totalBytes = (totalBits + 7) >> 3;
    for (u32 i = flushedWords * 4 + (u32)lane; i < totalBytes; i += ZSA_WAVE) wv_stg8(dst + i, ((const u8*)E->bitbuf)[i - flushedWords * 4]);
    wv_lds_fence();
    return totalBytes;
}

// ------------------------------------------------------------------ literals section (zstd_compress_literals.c:130-236)
// Writes the section at dst; returns its size (uniform).  `tmp` is global scratch of >= litSize + 64 bytes.
ZSA_DEV u32 zsa_noCompressLiterals(u8* dst, const u8* lit, u32 litSize, int lane)
{   // zstd_compress_literals.c:40-67
    u32 const flSize = 1 + (litSize > 31) + (litSize > 4095);
    if (lane == 0) {
        if (flSize == 1) dst[0] = (u8)(zsa_set_basic + (litSize << 3));
        else if (flSize == 2) zsa_st16(dst, zsa_set_basic + (1 << 2) + (litSize << 4));
        else zsa_st24(dst, zsa_set_basic + (3 << 2) + (litSize << 4));
    }
    for (u32 i = (u32)lane; i < litSize; i += ZSA_WAVE) dst[flSize + i] = lit[i];
    return flSize + litSize;
}
ZSA_DEV u32 zsa_compressLiterals(ZsaEncShared* S, u8* dst, const u8* lit, u32 litSize, u8* tmp,
                                 int disableLiteralCompression, int suspectUncompressible, u32 strategy, int lane)
{
    u32 const lhSize = 3 + (litSize >= 1024) + (litSize >= 16384);
    // nextHuf := prevHuf (:153)
    for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { S->hufNext.nbBits[i] = S->hufPrev.nbBits[i]; S->hufNext.val[i] = S->hufPrev.val[i]; }
    if (lane == 0) S->nextHufRepeat = S->prevHufRepeat;
    wv_sync();
    u32 const prevRepeat = S->prevHufRepeat;
    u32 const mintc = (prevRepeat == zsa_rep_valid) ? 6 : 64;      // :115-128: 8 << min(9 - strategy, 3), strategies fast .. lazy2
    if (disableLiteralCompression || litSize < mintc) return zsa_noCompressLiterals(dst, lit, litSize, lane);

    u32 singleStream = litSize < 256;
    if (prevRepeat == zsa_rep_valid && lhSize == 3) singleStream = 1;
    int const preferRepeat = (strategy < 4) && (litSize <= 1024);   // :166: strategy < ZSTD_lazy
    u8* const ostart = dst + lhSize;

    // ---- huf_compress.c:1298-1406 HUF_compress_internal
    // decision: 0 -> not compressible (raw), 1 -> rle, 2 -> encode with old table, 3 -> encode with new table (header written)
    u32 decision = 0xFF;
    if (preferRepeat && prevRepeat == zsa_rep_valid) decision = 2;
    if (decision == 0xFF && suspectUncompressible && litSize >= 4096 * 10) {
        u32 largestTotal = 0;
        zsa_hist_bytes(S, lit, 4096, lane);
        {   u32 m = 0; for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) m = S->count[i] > m ? S->count[i] : m; largestTotal = wv_reduce_max(m); }
        wv_sync();
        zsa_hist_bytes(S, lit + litSize - 4096, 4096, lane);
        {   u32 m = 0; for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) m = S->count[i] > m ? S->count[i] : m; largestTotal += wv_reduce_max(m); }
        wv_sync();
        if (largestTotal <= ((2 * 4096) >> 7) + 4) decision = 0;
    }
    u32 maxSV = 255;
    if (decision == 0xFF) {
        zsa_hist_bytes(S, lit, litSize, lane);
        u32 m = 0, top = 0;
        for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { u32 const c = S->count[i]; if (c > m) m = c; if (c) top = i; }
        u32 const largest = wv_reduce_max(m);
        maxSV = wv_reduce_max(top);
        if (largest == litSize) decision = 1;
        else if (largest <= (litSize >> 7) + 4) decision = 0;
    }
    if (decision == 0xFF) {
        {   // (all lanes, uniform)
            u32 repeat = prevRepeat, dec = 3, hSize = 0;
            if (repeat == zsa_rep_check) {   // HUF_validateCTable (:779-787)
                bool bad = false;
                for (u32 s = (u32)lane; s <= maxSV; s += ZSA_WAVE) bad |= (S->count[s] != 0) & (S->hufPrev.nbBits[s] == 0);
                if (wv_ballot(bad)) repeat = zsa_rep_none;
            }
            if (preferRepeat && repeat != zsa_rep_none) dec = 2;
            else {
                u32 huffLog = zsa_fse_optimalTableLog(11, litSize, maxSV, 1);
                u32 const maxBits = zsa_huf_buildCTable_wave(S, &S->hufNext, S->count, maxSV, huffLog, lane);
                if (maxBits == 0) dec = 0;   // error -> literals stored raw (zstd_compress_literals.c:189)
                else {
                    huffLog = maxBits;
                    hSize = zsa_huf_writeCTable(S, ostart, &S->hufNext, maxSV, huffLog, lane);
                    if (hSize == 0) dec = 0;
                    else {
                        if (repeat != zsa_rep_none) {
                            u32 const oldSize = zsa_huf_estimate(&S->hufPrev, S->count, maxSV, lane);
                            u32 const newSize = zsa_huf_estimate(&S->hufNext, S->count, maxSV, lane);
                            if (oldSize <= hSize + newSize || hSize + 12 >= litSize) dec = 2;
                        }
                        if (dec == 3 && hSize + 12 >= litSize) dec = 0;
                        if (dec == 3) repeat = zsa_rep_none;
                    }
                }
            }
            wv_sync();
            if (lane == 0) { S->u[EU_useOld] = dec; S->u[EU_hSize] = (dec == 3) ? hSize : 0; S->u[EU_tmp] = repeat; }
        }
        wv_sync();
        decision = S->u[EU_useOld];
    } else {
        if (lane == 0) { S->u[EU_hSize] = 0; S->u[EU_tmp] = prevRepeat; }
        wv_sync();
    }
    u32 const hSize = S->u[EU_hSize];
    u32 const repeatAfter = S->u[EU_tmp];
    ZSA_PROF(S, lane, 3);
    u32 cLitSize = 0;   // 0 -> raw
    if (decision == 1) cLitSize = 1;
    else if (decision >= 2) {
        const ZsaHufCT* const ct = (decision == 2) ? &S->hufPrev : &S->hufNext;
        u8* const op = ostart + hSize;
        u32 total = 0;
        // a literal section of litSize bytes or more is never kept (:1192-1207, zstd_compress_literals.c:186-191): that is the budget
        u32 const budget = litSize > hSize ? litSize - hSize : 0;
        if (singleStream) {
            total = zsa_huf_encode_wave(S, op, lit, litSize, ct, budget, lane);
            if (total == ZSA_HUF_NOFIT) total = 0;
        } else if (litSize >= 12) {
            // huf_compress.c:1135-1183: 6-byte jump table, 4 segments of (n+3)/4, written one after the other
            u32 const segmentSize = (litSize + 3) / 4;
            u32 c[4] = { ZSA_HUF_NOFIT, ZSA_HUF_NOFIT, ZSA_HUF_NOFIT, ZSA_HUF_NOFIT }, at = 6;
            for (u32 k = 0; k < 4; k++) {
                u32 const beg = k * segmentSize;
                u32 const len = (k == 3) ? litSize - 3 * segmentSize : segmentSize;
                c[k] = zsa_huf_encode_wave(S, op + at, lit + beg, len, ct, budget > at ? budget - at : 0, lane);
                if (c[k] == ZSA_HUF_NOFIT) break;
                at += c[k];
            }
            if (c[0] > 65535 || c[1] > 65535 || c[2] > 65535 || c[3] > 65535) total = 0;   // (includes ZSA_HUF_NOFIT)
            else {
                if (lane == 0) { zsa_st16(op, c[0]); zsa_st16(op + 2, c[1]); zsa_st16(op + 4, c[2]); }
                total = at;
            }
        }
        // huf_compress.c:1192-1207 HUF_compressCTable_internal
        if (total == 0) cLitSize = 0;
        else if (hSize + total >= litSize - 1) cLitSize = 0;
        else cLitSize = hSize + total;
    }
    wv_sync();
    u32 hType = zsa_set_compressed;
    if (decision >= 1 && repeatAfter != zsa_rep_none) hType = zsa_set_repeat;     // zstd_compress_literals.c:181-185
    if (decision == 0 && repeatAfter != zsa_rep_none) hType = zsa_set_repeat;
    {   u32 const minGain = (litSize >> 6) + 2;
        if (cLitSize == 0 || cLitSize >= litSize - minGain) {
            // restore nextHuf := prevHuf (:190)
            for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { S->hufNext.nbBits[i] = S->hufPrev.nbBits[i]; S->hufNext.val[i] = S->hufPrev.val[i]; }
            wv_sync();
            return zsa_noCompressLiterals(dst, lit, litSize, lane);
        }
    }
    if (cLitSize == 1) {   // :193-202 (litSize >= 64 here unless repeat-valid path; then check bytes)
        u32 identical = 1;
        if (litSize < 8) { for (u32 i = 1; i < litSize; i++) if (lit[i] != lit[0]) identical = 0; }
        if (litSize >= 8 || identical) {
            for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { S->hufNext.nbBits[i] = S->hufPrev.nbBits[i]; S->hufNext.val[i] = S->hufPrev.val[i]; }
            wv_sync();
            u32 const flSize = 1 + (litSize > 31) + (litSize > 4095);   // zstd_compress_literals.c:82-108
            if (lane == 0) {
                if (flSize == 1) dst[0] = (u8)(zsa_set_rle + (litSize << 3));
                else if (flSize == 2) zsa_st16(dst, zsa_set_rle + (1 << 2) + (litSize << 4));
                else zsa_st24(dst, zsa_set_rle + (3 << 2) + (litSize << 4));
                dst[flSize] = lit[0];
            }
            return flSize + 1;
        }
    }
    if (decision == 2) {
        // the old table stays: nextHuf must hold prevHuf (it was overwritten if a new table was built and rejected)
        for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { S->hufNext.nbBits[i] = S->hufPrev.nbBits[i]; S->hufNext.val[i] = S->hufPrev.val[i]; }
    }
    if (lane == 0) {
        if (hType == zsa_set_compressed) S->nextHufRepeat = zsa_rep_check;
        if (lhSize == 3) zsa_st24(dst, hType + ((u32)(!singleStream) << 2) + (litSize << 4) + (cLitSize << 14));
        else if (lhSize == 4) zsa_st32(dst, hType + (2 << 2) + (litSize << 4) + (cLitSize << 18));
        else { zsa_st32(dst, hType + (3 << 2) + (litSize << 4) + (cLitSize << 22)); dst[4] = (u8)(cLitSize >> 10); }
    }
    wv_sync();
    return lhSize + cLitSize;
}

// ------------------------------------------------------------------ sequences section
ZSA_DEV u32 zsa_LLcode(u32 ll)
{   // zstd_compress_internal.h:523-535
    if (ll > 63) return zsa_highbit32(ll) + 19;
    if (ll < 16) return ll;
    if (ll < 24) return 16 + ((ll - 16) >> 1);
    if (ll < 32) return 20 + ((ll - 24) >> 2);
    if (ll < 48) return 22 + ((ll - 32) >> 3);
    return 24;
}
ZSA_DEV u32 zsa_MLcode(u32 mlBase)
{   // zstd_compress_internal.h:540-552
    if (mlBase > 127) return zsa_highbit32(mlBase) + 36;
    if (mlBase < 32) return mlBase;
    if (mlBase < 40) return 32 + ((mlBase - 32) >> 1);
    if (mlBase < 48) return 36 + ((mlBase - 40) >> 2);
    if (mlBase < 64) return 38 + ((mlBase - 48) >> 3);
    if (mlBase < 96) return 40 + ((mlBase - 64) >> 4);
    return 42;
}
// zstd_compress_sequences.c:22-45 kInverseProbabilityLog256: floor(-log2(x / 256) * 256) for x in [1, 256), 0 for x == 0
ZSA_DEV u32 zsa_invProbLog256(u32 x)
{
    static const u16 t[256] = {
        0,    2048, 1792, 1642, 1536, 1453, 1386, 1329, 1280, 1236, 1197, 1162, 1130, 1100, 1073, 1047,
        1024, 1001, 980,  960,  941,  923,  906,  889,  874,  859,  844,  830,  817,  804,  791,  779,
        768,  756,  745,  734,  724,  714,  704,  694,  685,  676,  667,  658,  650,  642,  633,  626,
        618,  610,  603,  595,  588,  581,  574,  567,  561,  554,  548,  542,  535,  529,  523,  517,
        512,  506,  500,  495,  489,  484,  478,  473,  468,  463,  458,  453,  448,  443,  438,  434,
        429,  424,  420,  415,  411,  407,  402,  398,  394,  390,  386,  382,  377,  373,  370,  366,
        362,  358,  354,  350,  347,  343,  339,  336,  332,  329,  325,  322,  318,  315,  311,  308,
        305,  302,  298,  295,  292,  289,  286,  282,  279,  276,  273,  270,  267,  264,  261,  258,
        256,  253,  250,  247,  244,  241,  239,  236,  233,  230,  228,  225,  222,  220,  217,  215,
        212,  209,  207,  204,  202,  199,  197,  194,  192,  190,  187,  185,  182,  180,  178,  175,
        173,  171,  168,  166,  164,  162,  159,  157,  155,  153,  151,  149,  146,  144,  142,  140,
        138,  136,  134,  132,  130,  128,  126,  123,  121,  119,  117,  115,  114,  112,  110,  108,
        106,  104,  102,  100,  98,   96,   94,   93,   91,   89,   87,   85,   83,   82,   80,   78,
        76,   74,   73,   71,   69,   67,   66,   64,   62,   61,   59,   57,   55,   54,   52,   50,
        49,   47,   46,   44,   42,   41,   39,   37,   36,   34,   33,   31,   30,   28,   26,   25,
        23,   22,   20,   19,   17,   16,   14,   13,   11,   10,   8,    7,    5,    4,    2,    1 };
    return t[x];
}
#define ZSA_COST_ERROR 0xFFFFFFFFu   // ERROR(GENERIC) as a cost: above every real one
// zstd_compress_sequences.c:157-236 ZSTD_selectEncodingType, by the whole wave (lane s <-> code s; count: LDS; every argument and
// the result wave-uniform, *repeatMode a per-lane copy of a uniform value).  Below ZSTD_lazy: threshold rules (:180-205).  From
// ZSTD_lazy on: cost estimates (:206-232), each a sum over the codes and so a wave sum -- basic = cross entropy against the
// default distribution (:140-155), repeat = price under the previous block's table `prevCT` (:105-133 with fse.h:494-509;
// NULL when there is none), compressed = table description (:71-79, written to `wksp` just to be measured) + entropy
// bound (:85-99).
ZSA_DEV u32 zsa_selectEncodingType(ZsaEncShared* S, u32* repeatMode, const u32* count, u32 max, u32 mostFrequent, u32 nbSeq,
                                   int kind, const ZsaFseCT* prevCT, u8* wksp, int isDefaultAllowed, u32 strategy, int lane)
{
    u32 const defaultNormLog = kind == 1 ? ZSA_OF_DEFAULTNORMLOG : 6;
    if (mostFrequent == nbSeq) {
        *repeatMode = zsa_rep_none;
        if (isDefaultAllowed && nbSeq <= 2) return zsa_set_basic;
        return zsa_set_rle;
    }
    if (strategy < 4) {
        if (isDefaultAllowed) {
            u32 const dynamicFse_nbSeq_min = ((1u << defaultNormLog) * (10 - strategy)) >> 3;   // :185
            if ((*repeatMode == zsa_rep_valid) && (nbSeq < 1000)) return zsa_set_repeat;
            if ((nbSeq < dynamicFse_nbSeq_min) || (mostFrequent < (nbSeq >> (defaultNormLog - 1)))) {
                *repeatMode = zsa_rep_none;
                return zsa_set_basic;
            }
        }
    } else {
        u32 basicCost = ZSA_COST_ERROR, repeatCost = ZSA_COST_ERROR, compressedCost;
        bool const mine = (u32)lane <= max;
        u32 const c = mine ? count[lane] : 0;
        if (isDefaultAllowed) {
            u32 const shift = 8 - defaultNormLog;
            s32 const n = kind == 0 ? zsa_LL_defaultNorm((u32)lane) : (kind == 1 ? zsa_OF_defaultNorm((u32)lane) : zsa_ML_defaultNorm((u32)lane));
            u32 const normAcc = (n != -1) ? (u32)n : 1;
            basicCost = wv_reduce_add(mine ? c * zsa_invProbLog256(normAcc << shift) : 0u) >> 8;
        }
        if (*repeatMode != zsa_rep_none && prevCT && prevCT->maxSV >= max) {
            u32 const tableLog = prevCT->tableLog, badCost = (tableLog + 1) << 8;
            u32 bitCost = 0;
            if (c != 0) {
                u32 const dnb = prevCT->dnb[lane];
                u32 const minNbBits = dnb >> 16;
                u32 const threshold = (minNbBits + 1) << 16;
                u32 const deltaFromThreshold = threshold - (dnb + (1u << tableLog));
                bitCost = ((minNbBits + 1) << 8) - ((deltaFromThreshold << 8) >> tableLog);
            }
            bool const bad = wv_ballot(c != 0 && bitCost >= badCost) != 0;
            u32 const cost = wv_reduce_add(c * bitCost);
            if (!bad) repeatCost = cost >> 8;
        }
        {   u32 const FSELog = kind == 1 ? ZSA_OffFSELog : 9;
            u32 const tableLog = zsa_fse_optimalTableLog(FSELog, nbSeq, max, 2);
            u32 NCountCost = 0;
            if (zsa_fse_normalizeCount_wave(S->norm, tableLog, count, nbSeq, max, nbSeq >= 2048, lane)) NCountCost = ZSA_COST_ERROR >> 3;
            else {
                NCountCost = zsa_fse_writeNCount_wave(S, wksp, S->norm, max, tableLog, lane);
            }
            u32 norm = (256 * c) / nbSeq;
            if (c != 0 && norm == 0) norm = 1;
            u32 const cost = wv_reduce_add(c != 0 ? c * zsa_invProbLog256(norm) : 0u);
            compressedCost = (NCountCost << 3) + (cost >> 8);
        }
        if (basicCost <= repeatCost && basicCost <= compressedCost) { *repeatMode = zsa_rep_none; return zsa_set_basic; }
        if (repeatCost <= compressedCost) return zsa_set_repeat;
    }
    *repeatMode = zsa_rep_check;
    return zsa_set_compressed;
}
// zstd_compress_sequences.c:243-289 ZSTD_buildCTable, by the whole wave.  kind 0 LL, 1 OF, 2 ML.  ct, count: LDS.  Returns header
// bytes (wave-uniform), sets *fail.
ZSA_DEV u32 zsa_buildSeqCTable(ZsaEncShared* S, u8* dst, ZsaFseCT* ct, int kind, u32 type, u32* count, u32 max,
                               u32 lastCode, u32 firstCode, u32 nbSeq, const ZsaFseCT* dictTable, u32* fail, int lane)
{
    u32 const FSELog = kind == 1 ? ZSA_OffFSELog : 9;
    switch (type) {
    case zsa_set_rle:
        if (lane == 0) {
            ct->tableLog = 0; ct->maxSV = max; ct->nextState[0] = 0; ct->nextState[1] = 0; ct->dnb[max] = 0; ct->dfs[max] = 0;
            *dst = (u8)firstCode;
        }
        wv_lds_fence();
        return 1;
    case zsa_set_repeat:
        if (dictTable) {
            for (u32 i = (u32)lane; i < 512; i += ZSA_WAVE) ct->nextState[i] = dictTable->nextState[i];
            if ((u32)lane <= ZSA_MaxSeq) { ct->dfs[lane] = dictTable->dfs[lane]; ct->dnb[lane] = dictTable->dnb[lane]; }
            if (lane == 0) { ct->tableLog = dictTable->tableLog; ct->maxSV = dictTable->maxSV; }
            wv_lds_fence();
        }
        return 0;
    case zsa_set_basic: {
        u32 const dmax = kind == 0 ? ZSA_MaxLL : (kind == 1 ? ZSA_DefaultMaxOff : ZSA_MaxML);
        u32 const dlog = kind == 1 ? ZSA_OF_DEFAULTNORMLOG : 6;
        if ((u32)lane <= dmax)
            S->norm[lane] = (s16)(kind == 0 ? zsa_LL_defaultNorm((u32)lane) : (kind == 1 ? zsa_OF_defaultNorm((u32)lane) : zsa_ML_defaultNorm((u32)lane)));
        wv_lds_fence();
        zsa_fse_buildCTable_wave(S, ct->nextState, ct->dfs, ct->dnb, S->norm, dmax, dlog, lane);
        if (lane == 0) { ct->tableLog = dlog; ct->maxSV = dmax; }
        wv_lds_fence();
        return 0; }
    default: {
        u32 nbSeq_1 = nbSeq;
        u32 const tableLog = zsa_fse_optimalTableLog(FSELog, nbSeq, max, 2);
        u32 const cLast = count[lastCode];
        wv_lds_fence();
        if (cLast > 1) { if (lane == 0) count[lastCode] = cLast - 1; nbSeq_1--; }
        wv_lds_fence();
        if (zsa_fse_normalizeCount_wave(S->norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048, lane)) { *fail = 1; return 0; }
        u32 const NCountSize = zsa_fse_writeNCount_wave(S, dst, S->norm, max, tableLog, lane);
        if (NCountSize == 0) { *fail = 1; return 0; }
        zsa_fse_buildCTable_wave(S, ct->nextState, ct->dfs, ct->dnb, S->norm, max, tableLog, lane);
        if (lane == 0) { ct->tableLog = tableLog; ct->maxSV = max; }
        wv_lds_fence();
        return NCountSize; }
    }
}

// Entropy stage of one block: zstd_compress.c:2831-2986.  seq/nbSeq and the gathered literals are in global
// scratch.  Returns the block body size, 0 for "store raw"; S->err on internal failure.
ZSA_DEV u32 zsa_entropyCompress(ZsaEncShared* S, const u64* seq, u32 nbSeq, const u8* lit, u32 litSize,
                                u8* dst, u32 srcSize, u8* tmp, int literalsDisabled, const ZsaCDictDev* cd, u32 strategy,
                                const ZsaFseCT* park, int lane)
{
    u8* op = dst;
    {   int const suspect = (nbSeq == 0) || (litSize / nbSeq >= 20);
        op += zsa_compressLiterals(S, op, lit, litSize, tmp, literalsDisabled, suspect, strategy, lane);
    }
    wv_sync();
    ZSA_PROF(S, lane, 4);
    if (lane == 0) {
        u8* p = op;
        if (nbSeq < 128) *p++ = (u8)nbSeq;
        else if (nbSeq < ZSA_LONGNBSEQ) { p[0] = (u8)((nbSeq >> 8) + 0x80); p[1] = (u8)nbSeq; p += 2; }
        else { p[0] = 0xFF; zsa_st16(p + 1, nbSeq - ZSA_LONGNBSEQ); p += 3; }
        S->u[EU_tmp] = (u32)(p - op);
        S->nextLL = S->prevLL; S->nextOF = S->prevOF; S->nextML = S->prevML;
    }
    wv_sync();
    op += S->u[EU_tmp];
    if (nbSeq != 0) {
        u8* const seqHead = op++;
        u32 types[3], lastCountSize = 0;
        u64 const lastSeq = seq[nbSeq - 1], firstSeq = seq[0];
        for (int kind = 0; kind < 3; kind++) {
            // histogram of the codes of this kind (hist.c; ZSTD_seqToCodes zstd_compress.c:2637-2663)
            for (u32 i = (u32)lane; i <= ZSA_MaxSeq; i += ZSA_WAVE) S->count[i] = 0;
            wv_sync();
            for (u32 i = (u32)lane; i < nbSeq; i += ZSA_WAVE) {
                u64 const s = seq[i];
                u32 const c = kind == 0 ? zsa_LLcode(zsa_seq_ll(s)) : (kind == 1 ? zsa_highbit32(zsa_seq_of(s)) : zsa_MLcode(zsa_seq_ml(s)));
                wv_atomic_add_lds(&S->count[c], 1);
            }
            wv_sync();
            {   // the table of this kind: decided and built by the whole wave, lane s <-> code s
                u32 const cMine = (u32)lane <= ZSA_MaxSeq ? S->count[lane] : 0;
                u32 const max = 63u - (u32)__builtin_clzll(wv_ballot(cMine != 0));      // (nbSeq > 0: some code occurs)
                u32 const mostFrequent = wv_reduce_max(cMine);
                u32* const rm = kind == 0 ? &S->nextLL : (kind == 1 ? &S->nextOF : &S->nextML);
                u32 rmv = *rm;
                int const defaultAllowed = kind == 1 ? (max <= ZSA_DefaultMaxOff) : 1;
                // the table a "repeat" would use: the dictionary's, else (ZSTD_lazy and up) the parked one of the last
                // compressed block of this frame
                const ZsaFseCT* const dt = cd ? (kind == 0 ? &cd->ll : (kind == 1 ? &cd->of : &cd->ml)) : (park ? park + kind : (const ZsaFseCT*)0);
                wv_lds_fence();
                // (below ZSTD_lazy the choice never prices the previous table, :180-205; from ZSTD_lazy on it does -- with a dictionary that
                // is the dictionary's table: such frames have one block, the attach cut-off is 32 KiB)
                u32 const type = zsa_selectEncodingType(S, &rmv, S->count, max, mostFrequent, nbSeq, kind, (cd && strategy < 4) ? (const ZsaFseCT*)0 : dt, tmp, defaultAllowed, strategy, lane);
                u32 const lastCode = kind == 0 ? zsa_LLcode(zsa_seq_ll(lastSeq)) : (kind == 1 ? zsa_highbit32(zsa_seq_of(lastSeq)) : zsa_MLcode(zsa_seq_ml(lastSeq)));
                u32 const firstCode = kind == 0 ? zsa_LLcode(zsa_seq_ll(firstSeq)) : (kind == 1 ? zsa_highbit32(zsa_seq_of(firstSeq)) : zsa_MLcode(zsa_seq_ml(firstSeq)));
                ZsaFseCT* const ct = kind == 0 ? &S->ll : (kind == 1 ? &S->of : &S->ml);
                u32 fail = 0;
                u32 const cs = zsa_buildSeqCTable(S, op, ct, kind, type, S->count, max, lastCode, firstCode, nbSeq, dt, &fail, lane);
                if (lane == 0) {
                    *rm = rmv;
                    if (fail) S->err = ZSA_GENERIC;
                }
                wv_sync();
                if (S->err) return 0;
                types[kind] = type;
                if (type == zsa_set_compressed) lastCountSize = cs;
                op += cs;
            }
            wv_sync();
        }
        ZSA_PROF(S, lane, 5);
        // zstd_compress_sequences.c:291-383 ZSTD_encodeSequences_body.  The bitstream is written last sequence first; the
        // only serial parts are the three FSE state chains, which are independent of each other.  64 sequences at a
        // time: all lanes derive codes and extra bits, lanes 0/1/2 walk the OF/ML/LL state chains over the 64 codes,
        // all lanes assemble their sequence's bits, a prefix sum places them, LDS atomic-or packs them, and whole
        // words are flushed to the output together.
        if (lane == 0) *seqHead = (u8)((types[0] << 6) + (types[1] << 4) + (types[2] << 2));
        {
            ZsaSeqEncScratch* const E = &S->enc;
            for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) E->bitbuf[i] = 0;
            const ZsaFseCT* const myCT = lane == 0 ? &S->of : (lane == 1 ? &S->ml : &S->ll);
            ZsaFseCState st; st.value = 0; st.next = myCT->nextState; st.dfs = myCT->dfs; st.dnb = myCT->dnb; st.stateLog = myCT->tableLog;
            u32 totalBits = 0, flushedWords = 0;
            // the chain lanes' tables and batch arrays as LDS pointers (one table per lane: as generic pointers every access
            // of the chain paid a 64-bit addition and a null check, the last one of four a flat load)
            int const cl = lane < 3 ? lane : 0;
            ZSA_LDS const u8*  const cCodes = (ZSA_LDS const u8*)E->code[cl];
            ZSA_LDS const u16* const cNext = (ZSA_LDS const u16*)myCT->nextState;
            ZSA_LDS const s32* const cDfs = (ZSA_LDS const s32*)myCT->dfs;
            ZSA_LDS const u32* const cDnb = (ZSA_LDS const u32*)myCT->dnb;
            ZSA_LDS u16* const cBits = (ZSA_LDS u16*)E->stBits[cl];
            ZSA_LDS u8*  const cNb = (ZSA_LDS u8*)E->stNb[cl];
            wv_lds_fence();
            for (u32 tBase = 0; tBase < nbSeq; tBase += ZSA_WAVE) {
                u32 const cnt = (nbSeq - tBase) < ZSA_WAVE ? (nbSeq - tBase) : ZSA_WAVE;
                bool const mine = (u32)lane < cnt;
                u32 ll = 0, ml = 0, of = 1, llc = 0, mlc = 0, ofc = 0;
                if (mine) {
                    u64 const sq = seq[nbSeq - 1 - (tBase + (u32)lane)];
                    ll = zsa_seq_ll(sq); ml = zsa_seq_ml(sq); of = zsa_seq_of(sq);
                    llc = zsa_LLcode(ll); mlc = zsa_MLcode(ml); ofc = zsa_highbit32(of);
                    E->code[0][lane] = (u8)ofc; E->code[1][lane] = (u8)mlc; E->code[2][lane] = (u8)llc;
                }
                wv_lds_fence();
                if (lane < 3) {   // the three state chains, side by side
                    // One step = one table read that depends on the step before it.  What a step needs besides the state -- the
                    // symbol's two transform words -- depends on the symbol only, so it is fetched four steps at a time ahead
                    // of the chain: a step then costs one LDS round trip instead of four.
                    u32 j = 0;
                    u32 value = st.value;
#define ZSA_CHAIN_STEP(dnbv, dfsv, jj) do { u32 const nbBitsOut = (value + (dnbv)) >> 16;                                   \
                        cBits[jj] = (u16)(value & ((1u << nbBitsOut) - 1u)); cNb[jj] = (u8)nbBitsOut;      \
                        value = cNext[(s32)(value >> nbBitsOut) + (dfsv)]; } while (0)
                    if (tBase == 0) {   // FSE_initCState2 with the last sequence's code
                        zsa_fse_initCState2(st, st.next, st.dfs, st.dnb, st.stateLog, cCodes[0]);
                        value = st.value;
                        cBits[0] = 0; cNb[0] = 0;
                        j = 1;
                    }
                    for (; j + 4 <= cnt; j += 4) {
                        u32 const s0 = cCodes[j], s1 = cCodes[j + 1], s2 = cCodes[j + 2], s3 = cCodes[j + 3];
                        u32 const b0 = cDnb[s0], b1 = cDnb[s1], b2 = cDnb[s2], b3 = cDnb[s3];
                        s32 const f0 = cDfs[s0], f1 = cDfs[s1], f2 = cDfs[s2], f3 = cDfs[s3];
                        ZSA_CHAIN_STEP(b0, f0, j); ZSA_CHAIN_STEP(b1, f1, j + 1); ZSA_CHAIN_STEP(b2, f2, j + 2); ZSA_CHAIN_STEP(b3, f3, j + 3);
                    }
                    for (; j < cnt; j++) { u32 const sy = cCodes[j]; u32 const bv = cDnb[sy]; s32 const fv = cDfs[sy]; ZSA_CHAIN_STEP(bv, fv, j); }
                    st.value = value;
#undef ZSA_CHAIN_STEP
                }
                wv_lds_fence();
                // this sequence's bits, in emission order: OF/ML/LL state bits, then LL/ML/OF extra bits
                u64 lo = 0, hi = 0; u32 len = 0;
#define ZSA_BLOB_ADD(val, nb) do { u32 const nb_ = (nb); u64 const v_ = (u64)(val); if (nb_) { if (len < 64) { lo |= v_ << len; if (len + nb_ > 64) hi |= v_ >> (64 - len); } else hi |= v_ << (len - 64); len += nb_; } } while (0)
                if (mine) {
                    ZSA_BLOB_ADD(E->stBits[0][lane], E->stNb[0][lane]);
                    ZSA_BLOB_ADD(E->stBits[1][lane], E->stNb[1][lane]);
                    ZSA_BLOB_ADD(E->stBits[2][lane], E->stNb[2][lane]);
                    { u32 const b = zsa_LL_bits(llc); ZSA_BLOB_ADD(ll & ((1u << b) - 1u), b); }
                    { u32 const b = zsa_ML_bits(mlc); ZSA_BLOB_ADD(ml & ((1u << b) - 1u), b); }
                    ZSA_BLOB_ADD(of & ((1u << ofc) - 1u), ofc);
                }
                u32 const incl = wv_scan_incl(len);
                u32 const chunkBits = wv_readlane(incl, 63);
                if (mine && len) {
                    u32 const rel = totalBits + (incl - len) - flushedWords * 32;
                    u32 const w = rel >> 5, sh = rel & 31;
                    u64 const nlo = lo << sh, nhi = (hi << sh) | (sh ? (lo >> (64 - sh)) : 0);
                    if ((u32)nlo) wv_atomic_or_lds(&E->bitbuf[w], (u32)nlo);
                    if ((u32)(nlo >> 32)) wv_atomic_or_lds(&E->bitbuf[w + 1], (u32)(nlo >> 32));
                    if ((u32)nhi) wv_atomic_or_lds(&E->bitbuf[w + 2], (u32)nhi);
                    if ((u32)(nhi >> 32)) wv_atomic_or_lds(&E->bitbuf[w + 3], (u32)(nhi >> 32));
                }
                totalBits += chunkBits;
                wv_lds_fence();
                // a block body of srcSize bytes or more is never kept (:2975-2979; the reference's own writer stops at the end of
                // its buffer and the block then goes out raw, :2929-2932): give up as soon as that is known -- with tables taken
                // over from a dictionary the bitstream of many short sequences can outgrow the block
                if ((u32)(op - dst) + ((totalBits + 7) >> 3) > srcSize) return 0;
                {   // flush whole words, carry the partial one
                    u32 const done = (totalBits >> 5) - flushedWords;
                    for (u32 i = (u32)lane; i < done; i += ZSA_WAVE) wv_stg32(op + 4 * (flushedWords + i), E->bitbuf[i]);
                    u32 const carry = E->bitbuf[done];
                    wv_lds_fence();
                    for (u32 i = (u32)lane; i <= done + 4 && i < 256; i += ZSA_WAVE) E->bitbuf[i] = 0;
                    wv_lds_fence();
                    if (lane == 0) E->bitbuf[0] = carry;
                    flushedWords += done;
                    wv_lds_fence();
                }
            }
            // FSE_flushCState x3 (ML, OF, LL) + end mark (bitstream.h:229-241)
            u32 const vOF = wv_readlane(st.value, 0), vML = wv_readlane(st.value, 1), vLL = wv_readlane(st.value, 2);
            {   u64 lo = 0, hi = 0; u32 len = 0;
                ZSA_BLOB_ADD(vML & ((1u << S->ml.tableLog) - 1u), S->ml.tableLog);
                ZSA_BLOB_ADD(vOF & ((1u << S->of.tableLog) - 1u), S->of.tableLog);
                ZSA_BLOB_ADD(vLL & ((1u << S->ll.tableLog) - 1u), S->ll.tableLog);
                ZSA_BLOB_ADD(1u, 1);
                if (lane == 0) {
                    u32 const rel = totalBits - flushedWords * 32;
                    u32 const w = rel >> 5, sh = rel & 31;
                    u64 const nlo = lo << sh, nhi = sh ? (lo >> (64 - sh)) : 0;
                    E->bitbuf[w] |= (u32)nlo; E->bitbuf[w + 1] |= (u32)(nlo >> 32); E->bitbuf[w + 2] |= (u32)nhi;
                }
                totalBits += len;
            }
#undef ZSA_BLOB_ADD
            wv_lds_fence();
            u32 const totalBytes = (totalBits + 7) >> 3;
            for (u32 i = flushedWords * 4 + (u32)lane; i < totalBytes; i += ZSA_WAVE) wv_stg8(op + i, ((const u8*)E->bitbuf)[i - flushedWords * 4]);
            if (lane == 0) S->u[EU_tmp] = totalBytes;
        }
        wv_sync();
        u32 const bitstreamSize = S->u[EU_tmp];
        ZSA_PROF(S, lane, 6);
        op += bitstreamSize;
        wv_sync();
        if (lastCountSize && (lastCountSize + bitstreamSize) < 4) return 0;   // zstd_compress.c:2938-2944
    }
    {   u32 const cSize = (u32)(op - dst);
        u32 const maxCSize = srcSize - ((srcSize >> 6) + 2);
        if (cSize >= maxCSize) return 0;
        return cSize;
    }
}

// The same as a function of its own, for the kernels whose parsers are out-of-line functions too: the frame encoder then keeps
// only frame-level state alive across its calls instead of saving the entropy stages' hoisted values around every parser call
// (553 -> 82 spilled vector registers in the headline kernel; with 1 KiB payloads those saves were a quarter of the frame's time).
// The greedy/lazy kernel, whose parser is inlined, keeps the inlined form (measured: -13 % with this one).
ZSA_DEVFN u32 zsa_entropyCompress_fn(ZsaEncShared* S, const u64* seq, u32 nbSeq, const u8* lit, u32 litSize,
                                     u8* dst, u32 srcSize, u8* tmp, int literalsDisabled, const ZsaCDictDev* cd, u32 strategy)
{
    return zsa_entropyCompress(S, seq, nbSeq, lit, litSize, dst, srcSize, tmp, literalsDisabled, cd, strategy, (const ZsaFseCT*)0, wv_lane());
}

// gather the literal runs named by the recorded sequences into `lit` (all lanes), ZSTD_storeSeq's copy
// (zstd_compress_internal.h:675-687) + ZSTD_storeLastLiterals (zstd_compress.c:3065-3070)
ZSA_DEV void zsa_gatherLiterals(ZsaEncShared* S, const u64* seq, u32 nbSeq, const u8* src, u32 srcSize, u32 lastLL, u8* lit, int lane)
{
    u32 litPos = 0, srcPos = 0;
    for (u32 base = 0; base < nbSeq; base += ZSA_WAVE) {
        u32 const n = (nbSeq - base) < ZSA_WAVE ? (nbSeq - base) : ZSA_WAVE;
        u64 const s = (u32)lane < n ? seq[base + lane] : 0;
        u32 const ll = (u32)lane < n ? zsa_seq_ll(s) : 0;
        u32 const tot = (u32)lane < n ? ll + zsa_seq_ml(s) + ZSA_MINMATCH : 0;
        u32 const litIncl = wv_scan_incl(ll), totIncl = wv_scan_incl(tot);
        u32 const totalLit = wv_shfl(litIncl, 63), totalSrc = wv_shfl(totIncl, 63);
        // Runs of up to 32 bytes (nearly all of them on text-like data) are moved by the lane that owns the sequence, 8 bytes
        // per instruction; the longer ones are flattened over the lanes in units of 8 bytes (the last unit of a run is moved
        // back to end with the run).
        if ((u32)lane < n && ll <= 32) zsa_copy_run32(lit + litPos + (litIncl - ll), src + srcPos + (totIncl - tot), ll);
        if (wv_ballot(ll > 32)) {
            u32 const units = ll > 32 ? (ll + 7) >> 3 : 0;
            u32 const uIncl = wv_scan_incl(units);
            S->bLit[lane] = uIncl - units;
            S->bSrc[lane] = totIncl - tot; S->bLen[lane] = ll; S->bDst[lane] = litIncl - ll;
            u32 const totalUnits = wv_shfl(uIncl, 63);
            wv_lds_fence();
            for (u32 t = (u32)lane; t < totalUnits; t += ZSA_WAVE) {
                u32 lo = 0, hi = n - 1;
                while (lo < hi) { u32 const mid = (lo + hi + 1) >> 1; if (S->bLit[mid] <= t) lo = mid; else hi = mid - 1; }
                u32 const runLL = S->bLen[lo], runLit = S->bDst[lo];
                u32 b = 8u * (t - S->bLit[lo]);
                if (runLL - b < 8u) b = runLL - 8u;   // (runLL > 32)
                wv_stg64u(lit + litPos + runLit + b, wv_ldg64(src + srcPos + S->bSrc[lo] + b));
            }
            wv_lds_fence();
        }
        litPos += totalLit; srcPos += totalSrc;
    }
    for (u32 t = (u32)lane; t < lastLL; t += ZSA_WAVE) lit[litPos + t] = src[srcSize - lastLL + t];
}

// ------------------------------------------------------------------ the frame (zstd_compress.c:4392-4632, 5107-5144)
// hashTable: global memory (per-workgroup scratch, L2-resident), (1 << P.hlog) entries.  scratch: seq (u64 x (blockSizeMax/4 + 8)), lit (128 KiB + 64), tmp (lit-sized x 1.5 + 128).
// Returns bytes written or (u64)-error (uniform).
struct ZsaEncScratch { u64* seq; u8* lit; u8* tmp; ZsaFseCT* park; };   // park: 3 sequence tables (LL, OF, ML) of the last compressed block

// FAMILY 0: ZSTD_fast with a dictionary; FAMILY 1: ZSTD_greedy / ZSTD_lazy / ZSTD_lazy2; FAMILY 2: ZSTD_fast without a
// dictionary (levels 1, 2 and below: the headline path); FAMILY 3: ZSTD_dfast; FAMILY 4: ZSTD_dfast .. ZSTD_lazy2 with an attached
// dictionary (BulkProcessor at levels 3 and up, payloads up to the attach cut-off).  Each is compiled into a kernel of its own:
// each parser then gets the whole register budget (one kernel holding two of them spilled).
template <int FAMILY>
ZSA_DEV u64 zsa_encode_frame(ZsaEncShared* S, u32* hashTable, const ZsaEncParams& P, const u8* src, u64 srcSize64,
                             u8* dst, u64 dstCapacity, const ZsaEncScratch& W, const ZsaCDictDev* cd, int dictMode)
{
    int const lane = wv_lane();
    if (srcSize64 > 0xE0000000ull) return ZSA_ERR(ZSA_srcSize_wrong);
    u32 const srcSize = (u32)srcSize64;
    {   // capacity contract of this path: every reference call site passes ZSTD_compressBound(srcSize) (zstd.go:96-101)
        u64 const bound = (u64)srcSize + (srcSize >> 8) + ((srcSize < (128u << 10)) ? (((128u << 10) - srcSize) >> 11) : 0);
        if (dstCapacity < bound || dstCapacity < 18) return ZSA_ERR(ZSA_dstSize_tooSmall);
    }
    // frame header, zstd_compress.c:4474-4520
    u32 pos = 0;
    u32 const dictID = cd ? cd->dictID : 0;
    {
        u32 const dictIDSizeCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
        u32 const singleSegment = ((u64)1 << P.wlog) >= srcSize;
        u32 const fcsCode = (srcSize >= 256) + (srcSize >= 65536 + 256);
        pos = 5 + !singleSegment + (dictIDSizeCode == 3 ? 4 : dictIDSizeCode) + (fcsCode == 0 ? singleSegment : (1u << fcsCode));
        if (lane == 0) {
            u32 q = 4;
            zsa_st32(dst, ZSA_MAGIC);
            dst[q++] = (u8)(dictIDSizeCode + (singleSegment << 5) + (fcsCode << 6));
            if (!singleSegment) dst[q++] = (u8)((P.wlog - ZSA_WLOG_ABSMIN) << 3);
            if (dictIDSizeCode == 1) { dst[q++] = (u8)dictID; }
            else if (dictIDSizeCode == 2) { zsa_st16(dst + q, dictID); q += 2; }
            else if (dictIDSizeCode == 3) { zsa_st32(dst + q, dictID); q += 4; }
            if (fcsCode == 0) { if (singleSegment) dst[q++] = (u8)srcSize; }
            else if (fcsCode == 1) { zsa_st16(dst + q, srcSize - 256); q += 2; }
            else { zsa_st32(dst + q, srcSize); q += 4; }
        }
    }
    u8* op = dst + pos;
    if (srcSize == 0) {   // zstd_compress.c:5125-5132
        if (lane == 0) zsa_st24(op, 1);
        wv_sync();
        return pos + 3;
    }
    // per-frame state (zstd_compress.c:2045-2225, 1863-1872; with a dictionary :2270-2329, :2347-2426, :5027-5044)
    ZsaWindow Wn;
    Wn.base = src - 2; Wn.dictBase = src - 2; Wn.dictLimit = 2; Wn.lowLimit = 2; Wn.loadedDictEnd = 0; Wn.dms = nullptr;
    if (cd && dictMode == zsa_dm_copy) {   // ZSTD_copyCDictTableIntoCCtx: tags removed
        for (u32 i = (u32)lane; i < (1u << P.hlog); i += ZSA_WAVE) hashTable[i] = cd->hashTable[i] >> 8;
    } else {
        // ZSTD_dfast: the short-hash ("chain") table follows the long one in the same scratch region; greedy/lazy/lazy2: the
        // chain table, or the row finder's tag table (one byte per cell)
        u32 const cells = (1u << P.hlog) + (FAMILY == 4 ? (P.strat < 2 ? 0u : ((cd->useRow ? 4u << P.hlog : 4u << P.clog))) : zsa_auxTableBytes(P.strat, P.wlog, P.hlog, P.clog)) / 4u;
        for (u32 i = 4u * (u32)lane; i < cells; i += 4u * ZSA_WAVE) { hashTable[i] = 0; hashTable[i + 1] = 0; hashTable[i + 2] = 0; hashTable[i + 3] = 0; }
    }
    if (cd) {
        for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { S->hufPrev.nbBits[i] = cd->huf.nbBits[i]; S->hufPrev.val[i] = cd->huf.val[i]; }
    } else {
        for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { S->hufPrev.nbBits[i] = 0; S->hufPrev.val[i] = 0; }
    }
    if (lane == 0) {
        if (cd) {
            S->prevRep[0] = cd->rep[0]; S->prevRep[1] = cd->rep[1]; S->prevRep[2] = cd->rep[2];
            S->prevHufRepeat = cd->hufRepeat; S->prevLL = cd->llRepeat; S->prevOF = cd->ofRepeat; S->prevML = cd->mlRepeat;
        } else {
            S->prevRep[0] = 1; S->prevRep[1] = 4; S->prevRep[2] = 8;
            S->prevHufRepeat = zsa_rep_none; S->prevLL = S->prevOF = S->prevML = zsa_rep_none;
        }
        S->err = 0;
    }
    wv_sync();
    if (cd) {
        u32 const cdictEnd = 2 + cd->contentSize, cdictLen = cd->contentSize;
        if (dictMode == zsa_dm_attach) {
            if (cdictLen != 0) {
                Wn.dms = cd;
                Wn.base = src - cdictEnd; Wn.dictBase = Wn.base;
                Wn.dictLimit = cdictEnd; Wn.lowLimit = cdictEnd; Wn.loadedDictEnd = cdictEnd;
            }
        } else {
            if (dictMode == zsa_dm_load && cd->isLoaded && cdictLen > 8) {
                // zstd_compress.c:4740-4761 + zstd_fast.c:50-82 (dtlm_fast): every third position of the indexable
                // suffix is inserted, later positions overwrite earlier ones == the largest index wins per cell
                u32 const m = P.hlog;   // chainLog <= hashLog in every ZSTD_fast row
                u32 const maxDictSize = 8u << (m < 28 ? m : 28);
                u32 const from = cdictLen > maxDictSize ? cdictEnd - maxDictSize : 2;
                if (cdictEnd - from > 8) {
                    const u8* const dbase = cd->content - 2;
                    // loop bound: ip + 3 < (end - 8) + 2  <=>  index + 9 < cdictEnd
                    for (u32 idx = from + 3 * (u32)lane; idx + 9 < cdictEnd; idx += 3 * ZSA_WAVE)
                        wv_atomic_max_global(&hashTable[zsa_hash(zsa_ld64(dbase + idx), P.hlog, P.mml)], idx);
                }
                wv_device_fence();
                wv_sync();
            }
            if (cdictLen != 0) {
                Wn.dictBase = cd->content - 2;
                Wn.base = src - cdictEnd;
                Wn.lowLimit = 2; Wn.dictLimit = cdictEnd;
                if (Wn.dictLimit - Wn.lowLimit < 8) Wn.lowLimit = Wn.dictLimit;
            }
            Wn.loadedDictEnd = cd->isLoaded ? cdictEnd : 0;
        }
    }
    u32 blockSize;
    {   u64 const ws0 = ((u64)1 << P.wlog) < srcSize ? ((u64)1 << P.wlog) : srcSize;
        blockSize = ws0 < ZSA_BLOCKSIZE_MAX ? (u32)ws0 : ZSA_BLOCKSIZE_MAX;
    }
    // Without a dictionary the match table is touched by the wave parser only, which then keeps a fingerprint of the 4
    // bytes at each position in the cell bits the position does not need (zsa_parse.h); indices run up to srcSize + 2.
    u32 fpPosBits = 0;
    if (!cd && P.strat == 1) { u32 const bits = 32u - (u32)__builtin_clz(srcSize + 2u); if (bits <= 26u) fpPosBits = bits; }
    // chunk path (zsa_parse.h): levels 1-2 (step 2) with a table the LDS filter covers
    // ZSTD_dfast cells carry fingerprints as well (zsa_dfast.h) while the frame's positions leave at least 6 bits for them
    u32 dfPosBits = 0;
    if (FAMILY == 3) { u32 const bits = 32u - (u32)__builtin_clz(srcSize + 2u); if (bits <= 26u) dfPosBits = bits; }
    bool const chunkParse = FAMILY == 2 && fpPosBits != 0 && P.tlen <= 1 && P.hlog <= ZSA_FILT_HLOG_MAX + 1;
    int const literalsDisabled = (P.strat == 1) && (P.tlen > 0);   // zstd_compress_internal.h:605-618
    u32 const maxDist = 1u << P.wlog;
    ZsaLazyCtx LZ;   // greedy / lazy / lazy2 (zsa_lazy.h): match-finder state that lives across the blocks of the frame
    LZ.S = S; LZ.hashTable = hashTable; LZ.chainTable = hashTable + (1u << P.hlog);
    LZ.base = Wn.base; LZ.hlog = P.hlog; LZ.clog = P.clog; LZ.wlog = P.wlog; LZ.slog = P.slog;
    LZ.mls = P.mml < 4 ? 4 : (P.mml > 6 ? 6 : P.mml);
    LZ.rowLog = P.slog < 4 ? 4 : (P.slog > 6 ? 6 : P.slog);
    LZ.rowHashLog = P.hlog - LZ.rowLog;                             // zstd_compress.c:1991-1995
    LZ.useRow = FAMILY == 4 ? (cd->useRow != 0 && P.strat >= 3) : zsa_useRowFinder(P.strat, P.wlog);   // zstd_compress.c:2294 "cdict overrides"
    LZ.lowLimit = Wn.lowLimit; LZ.nextToUpdate = Wn.dictLimit;      // zstd_compress.c:4596-4600
    LZ.lazySkipping = 0;
    LZ.dms = nullptr; LZ.dmsDelta = 0; LZ.V.base = Wn.base; LZ.V.dbase = Wn.base; LZ.V.prefix = Wn.dictLimit; LZ.V.dictLow = Wn.dictLimit;
    if (FAMILY == 4 && Wn.dms) {
        LZ.dms = Wn.dms;
        LZ.dmsDelta = Wn.dictLimit - (2u + Wn.dms->contentSize);
        LZ.V.dbase = (Wn.dms->content - 2) - LZ.dmsDelta; LZ.V.dictLow = 2u + LZ.dmsDelta;
    }
    u32 tablesBuilt = 0;
    u32 remaining = srcSize;
    const u8* ip = src;
    int isFirstBlock = 1;
    while (remaining) {
        u32 const lastBlock = (blockSize >= remaining);
        if (remaining < blockSize) blockSize = remaining;
        {   // ZSTD_checkDictValidity (zstd_compress_internal.h:1199-1230)
            u32 const blockEndIdx = (u32)(ip + blockSize - Wn.base);
            if (blockEndIdx > Wn.loadedDictEnd + maxDist || Wn.loadedDictEnd != Wn.dictLimit) { Wn.loadedDictEnd = 0; Wn.dms = nullptr; }
        }
        {   // ZSTD_window_enforceMaxDist (zstd_compress_internal.h:1153-1190)
            u32 const blockStartIdx = (u32)(ip - Wn.base);
            if (blockStartIdx > maxDist + Wn.loadedDictEnd) {
                u32 const nl = blockStartIdx - maxDist;
                if (Wn.lowLimit < nl) Wn.lowLimit = nl;
                if (Wn.dictLimit < Wn.lowLimit) Wn.dictLimit = Wn.lowLimit;
                Wn.loadedDictEnd = 0; Wn.dms = nullptr;
            }
        }
        if (LZ.nextToUpdate < Wn.lowLimit) LZ.nextToUpdate = Wn.lowLimit;   // zstd_compress.c:4427
        LZ.lowLimit = Wn.lowLimit;
        u32 cSize = 0;
        tablesBuilt = 0;
        ZSA_PROF(S, lane, 0);
        if (blockSize >= 7) {   // zstd_compress.c:3156
            {   // :3174-3181 limited update after a very long match
                u32 const curr = (u32)(ip - Wn.base);
                if (curr > LZ.nextToUpdate + 384) { u32 const gap = curr - LZ.nextToUpdate - 384; LZ.nextToUpdate = curr - (gap < 192 ? gap : 192); }
            }
            bool const dictParse = FAMILY == 0 && ((Wn.lowLimit < Wn.dictLimit) || (Wn.dms != nullptr));
            if (dictParse && !(Wn.lowLimit < Wn.dictLimit)) {   // attached dictionary: wave-level parse (zsa_parse_dict.h)
                // its bitmap takes LDS of the entropy stages: the Huffman table at hand (the dictionary's, for the first block)
                // waits in global scratch meanwhile
                u32* const park = (u32*)W.tmp;
                for (u32 i = (u32)lane; i < sizeof(ZsaHufCT) / 4; i += ZSA_WAVE) park[i] = ((const u32*)&S->hufPrev)[i];
                wv_sync();
                for (u32 i = (u32)lane; i < sizeof(S->dupBits) / 4; i += ZSA_WAVE) S->dupBits[i] = 0;
                wv_sync();
                if (lane == 0) { S->nextRep[0] = S->prevRep[0]; S->nextRep[1] = S->prevRep[1]; S->nextRep[2] = S->prevRep[2]; }
                wv_sync();
                u32* const io = &S->u[EU_s0];   // offsets in and out, last literal run out
                if (lane == 0) { io[0] = S->prevRep[0]; io[1] = S->prevRep[1]; io[2] = 0; }
                wv_sync();
                u32 const nbSeq = wv_uniform(zsa_parse_dms_wave(hashTable, P.hlog, P.mml, P.tlen, Wn.base, Wn.dictLimit, Wn.dms, ip, blockSize, io, W.seq, S->dupBits));
                wv_sync();
                for (u32 i = (u32)lane; i < sizeof(ZsaHufCT) / 4; i += ZSA_WAVE) ((u32*)&S->hufPrev)[i] = park[i];
                if (lane == 0) {
                    S->nextRep[0] = io[0]; S->nextRep[1] = io[1];
                    S->u[EU_nbSeq] = nbSeq; S->u[EU_lastLL] = io[2];
                }
            } else if (dictParse) {   // dictionary as an external segment (payloads above 8 KiB with a dictionary): wave-level too
                for (u32 i = (u32)lane; i < sizeof(S->dupBits) / 4; i += ZSA_WAVE) S->dupBits[i] = 0;
                u32* const io = &S->u[EU_s0];
                if (lane == 0) { io[0] = S->prevRep[0]; io[1] = S->prevRep[1]; io[2] = 0; }
                wv_sync();
                u32 const nbSeq = wv_uniform(zsa_parse_ext_wave(hashTable, P.hlog, P.mml, P.tlen, P.wlog, Wn.base, Wn.dictBase, Wn.dictLimit, Wn.lowLimit,
                                                                Wn.loadedDictEnd, ip, blockSize, io, W.seq, S->dupBits));   // 0xFFFFFFFF: segment invalidated
                wv_sync();
                if (lane == 0) {
                    S->nextRep[0] = io[0]; S->nextRep[1] = io[1]; S->nextRep[2] = S->prevRep[2];
                    S->u[EU_nbSeq] = nbSeq; S->u[EU_lastLL] = io[2];
                }
            }
            wv_sync();
            if (!dictParse || S->u[EU_nbSeq] == 0xFFFFFFFFu) {
                wv_sync();
                u32 rep[3] = { S->prevRep[0], S->prevRep[1], S->prevRep[2] };
                u32 lastLL = 0, nbSeq;
                if (FAMILY == 4 && Wn.dms) {   // attached dictionary above ZSTD_fast (one block: payloads up to the attach cut-off)
                    if (P.strat == 2) {
                        u32* const io = &S->u[EU_s0];
                        if (lane == 0) { io[0] = rep[0]; io[1] = rep[1]; io[2] = 0; }
                        wv_sync();
                        nbSeq = wv_uniform(zsa_parse_dfast_dms(hashTable, hashTable + (1u << P.hlog), P.hlog, P.clog, P.mml, Wn.base, Wn.dictLimit, Wn.dms,
                                                               ip, blockSize, io, W.seq));
                        wv_sync();
                        rep[0] = io[0]; rep[1] = io[1]; lastLL = io[2];
                    } else {
                        nbSeq = zsa_parse_lazy<true>(LZ, Wn.dictLimit, ip, blockSize, P.strat - 3, rep, W.seq, &lastLL, lane);
                    }
                } else if (FAMILY == 1 || (FAMILY == 4 && P.strat >= 3)) {   // greedy / lazy / lazy2 without a dictionary segment: wave-uniform, zsa_lazy.h
                    // the greedy chunk path's row bitmap lies in `dupBits`, which overlays the last block's FSE tables (parked in
                    // global scratch from ZSTD_lazy on, never used again below it), not the Huffman table carried from block to block
                    if (P.strat == 3) for (u32 i = (u32)lane; i < sizeof(S->dupBits) / 4; i += ZSA_WAVE) S->dupBits[i] = 0;
                    wv_sync();
                    nbSeq = zsa_parse_lazy<false>(LZ, Wn.dictLimit, ip, blockSize, P.strat - 3, rep, W.seq, &lastLL, lane);
                    ZSA_SEQ_HOOK(W.seq, nbSeq);
                } else if (FAMILY == 3 || FAMILY == 4) {   // ZSTD_dfast without a dictionary segment: wave-level, zsa_dfast.h
                    // its two cell bitmaps lie in `filt`, which overlays table-construction scratch and the last block's FSE tables
                    // (never used again below ZSTD_lazy), not the Huffman table that is carried from block to block
                    for (u32 i = (u32)lane; i < sizeof(S->filt) / 4; i += ZSA_WAVE) S->filt[i] = 0;
                    u32* const io = &S->u[EU_s0];   // offsets in and out, last literal run out
                    if (lane == 0) { io[0] = rep[0]; io[1] = rep[1]; io[2] = 0; }
                    wv_sync();
                    nbSeq = wv_uniform(zsa_parse_dfast_wave(hashTable, hashTable + (1u << P.hlog), P.hlog, P.clog, P.mml, P.wlog, Wn.base, Wn.dictLimit,
                                                            ip, blockSize, io, W.seq, S->filt, S->filt + 512, dfPosBits));
                    wv_sync();
                    rep[0] = io[0]; rep[1] = io[1]; lastLL = io[2];
                } else {   // ZSTD_fast without a dictionary segment: the wave parser (zsa_parse.h)
                    // its stage takes the whole LDS: the previous block's Huffman table waits in global scratch meanwhile
                    u32* const park = (u32*)W.tmp;
                    for (u32 i = (u32)lane; i < sizeof(ZsaHufCT) / 4; i += ZSA_WAVE) park[i] = ((const u32*)&S->hufPrev)[i];
                    wv_sync();
                    if (chunkParse) {   // level 1 / 2: chunk path with the table's filter in LDS
                        u32 const cells = 1u << P.hlog;
                        u32 const fb = P.hlog <= ZSA_FILT_HLOG_MAX ? 4u : 2u, per = 32u / fb;   // filter bits per cell, cells per word
                        if (isFirstBlock) { for (u32 i = (u32)lane; i < cells / per; i += ZSA_WAVE) S->filt[i] = 0; }
                        else {   // the filter mirrors the top fb bits of every cell
                            for (u32 i = (u32)lane; i < cells / per; i += ZSA_WAVE) {
                                u32 w = 0;
                                for (u32 k = 0; k < per; k++) w |= (hashTable[per * i + k] >> (32u - fb)) << (fb * k);
                                S->filt[i] = w;
                            }
                        }
                        for (u32 i = (u32)lane; i < sizeof(S->dupBits) / 4; i += ZSA_WAVE) S->dupBits[i] = 0;
                        wv_sync();
                        nbSeq = zsa_parse_fast_block(hashTable, S->filt, S->dupBits, S->ring, S->tags, P, Wn.base, Wn.dictLimit, ip, blockSize, rep, W.seq, &lastLL, fpPosBits, false, lane);
                    } else {
                        // (a dictionary loaded contiguously below the block -- extDict too small to keep -- only widens the prefix)
                        nbSeq = zsa_parse_fast_block(hashTable, nullptr, nullptr, S->ring, S->tags, P, Wn.base, Wn.dictLimit, ip, blockSize, rep, W.seq, &lastLL, fpPosBits, Wn.loadedDictEnd != 0, lane);
                    }
                    wv_sync();
                    for (u32 i = (u32)lane; i < sizeof(ZsaHufCT) / 4; i += ZSA_WAVE) ((u32*)&S->hufPrev)[i] = park[i];
                }
                wv_sync();
                if (lane == 0) {
                    S->nextRep[0] = rep[0]; S->nextRep[1] = rep[1]; S->nextRep[2] = rep[2];
                    S->u[EU_nbSeq] = nbSeq; S->u[EU_lastLL] = lastLL;
                }
            }
            wv_sync();
            ZSA_PROF(S, lane, 1);
            u32 const nbSeq = S->u[EU_nbSeq], lastLL = S->u[EU_lastLL];
            // literal bytes = block bytes not covered by matches
            u32 litSize;
            {   u32 m = 0;
                for (u32 i = (u32)lane; i < nbSeq; i += ZSA_WAVE) m += zsa_seq_ml(W.seq[i]) + ZSA_MINMATCH;
                litSize = blockSize - wv_reduce_add(m);
            }
            zsa_gatherLiterals(S, W.seq, nbSeq, ip, blockSize, lastLL, W.lit, lane);
            wv_sync();
            ZSA_PROF(S, lane, 2);
            // (FAMILY 0: the strategy is 1 or 2, which folds the cost-based mode selection out of that kernel)
            u32 const strat = (FAMILY == 2 || FAMILY == 0) ? 1u : (FAMILY == 3 ? 2u : P.strat);
            if (FAMILY == 1 || FAMILY == 4) cSize = zsa_entropyCompress(S, W.seq, nbSeq, W.lit, litSize, op + 3, blockSize, W.tmp, literalsDisabled, cd, strat,
                                                         (P.strat >= 4 && !cd) ? W.park : (const ZsaFseCT*)0, lane);
            else cSize = wv_uniform(zsa_entropyCompress_fn(S, W.seq, nbSeq, W.lit, litSize, op + 3, blockSize, W.tmp, literalsDisabled, cd, strat));
            if (S->err) return ZSA_ERR(S->err);
            tablesBuilt = nbSeq != 0;
            if (!isFirstBlock && cSize < 25) {   // zstd_compress.c:4258-4269 (ZSTD_isRLE :3413-3433)
                u32 diff = 0;
                u8 const b0 = ip[0];
                for (u32 i = (u32)lane; i < blockSize; i += ZSA_WAVE) diff |= (ip[i] != b0);
                if (!wv_ballot(diff != 0)) { cSize = 1; if (lane == 0) op[3] = b0; }
            }
        }
        wv_sync();
        if (cSize > 1 && lane == 0) {   // zstd_compress.c:4272-4274 confirm repcodes + entropy tables
            S->prevRep[0] = S->nextRep[0]; S->prevRep[1] = S->nextRep[1]; S->prevRep[2] = S->nextRep[2];
            S->prevHufRepeat = S->nextHufRepeat; S->prevLL = S->nextLL; S->prevOF = S->nextOF; S->prevML = S->nextML;
        }
        if (cSize > 1) for (u32 i = (u32)lane; i < 256; i += ZSA_WAVE) { S->hufPrev.nbBits[i] = S->hufNext.nbBits[i]; S->hufPrev.val[i] = S->hufNext.val[i]; }
        if (FAMILY == 1 && cSize > 1 && tablesBuilt && P.strat >= 4 && !cd && !lastBlock) {
            // a later block may price or repeat these tables (zstd_compress_sequences.c:206-232): park them before the next
            // parse reuses their LDS.  A block without sequences leaves the parked ones in place (zstd_compress.c:2901-2906).
            const u32* const from[3] = { (const u32*)&S->ll, (const u32*)&S->of, (const u32*)&S->ml };
            for (int k = 0; k < 3; k++)
                for (u32 i = (u32)lane; i < sizeof(ZsaFseCT) / 4; i += ZSA_WAVE) ((u32*)(W.park + k))[i] = from[k][i];
        }
        wv_sync();
        if (lane == 0 && S->prevOF == zsa_rep_valid) S->prevOF = zsa_rep_check;   // :4279-4280
        if (cSize == 0) {   // ZSTD_noCompressBlock, zstd_compress_internal.h:569-579
            if (lane == 0) zsa_st24(op, lastBlock + ((u32)zsa_bt_raw << 1) + (blockSize << 3));
            for (u32 i = (u32)lane; i < blockSize; i += ZSA_WAVE) op[3 + i] = ip[i];
            cSize = 3 + blockSize;
        } else {
            if (lane == 0) zsa_st24(op, cSize == 1 ? lastBlock + ((u32)zsa_bt_rle << 1) + (blockSize << 3)
                                                   : lastBlock + ((u32)zsa_bt_compressed << 1) + (cSize << 3));
            cSize += 3;
        }
        ip += blockSize; remaining -= blockSize;
        op += cSize;
        isFirstBlock = 0;
        wv_sync();
        ZSA_PROF(S, lane, 7);
    }
    return (u64)(op - dst);
}
