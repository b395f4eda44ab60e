/* zo_entropy_enc.c -- oracle: histogram, FSE and Huffman encoders, literal + sequence sections.
 * TEST INFRASTRUCTURE ONLY (see zo_common.h).  Restates, in plain C, the arithmetic of the reference's
 * hist.c, fse_compress.c, huf_compress.c, zstd_compress_literals.c, zstd_compress_sequences.c and the
 * entropy driver in zstd_compress.c.  Output buffers are assumed >= ZSTD_compressBound(srcSize), which is
 * what every Go call site passes (zstd.go:96-101, zstd_bulk.go:64-69); capacity-failure paths that can
 * only trigger with smaller buffers are not restated.
 */
#include "zo_common.h"
#include "zo_compress.h"

const u8 zo_LL_bits[ZO_MaxLL + 1] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
const u8 zo_ML_bits[ZO_MaxML + 1] = { 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,
                                      1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
/* default distributions of the format (zstd_internal.h:131-169) */
const s16 zo_LL_defaultNorm[ZO_MaxLL + 1] = { 4,3,2,2,2,2,2,2, 2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
const s16 zo_ML_defaultNorm[ZO_MaxML + 1] = { 1,4,3,2,2,2,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,
                                              1,1,1,1,1,1,1,1, 1,1,1,1,1,1,-1,-1, -1,-1,-1,-1,-1 };
const s16 zo_OF_defaultNorm[ZO_DefaultMaxOff + 1] = { 1,1,1,1,1,1,2,2, 2,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1, -1,-1,-1,-1,-1 };

/* ------------------------------------------------------------------ histogram */
/* hist.c:30-55 HIST_count_simple / :141-165 HIST_count_wksp: exact counts, *maxSVPtr := largest present
 * symbol, returns the largest count.  (The 4-way parallel variant at :67-134 gives identical results.) */
u32 zo_hist(u32* count, u32* maxSVPtr, const u8* src, size_t n)
{
    u32 maxSV = *maxSVPtr, largest = 0, s;
    size_t i;
    memset(count, 0, (maxSV + 1) * sizeof(u32));
    if (n == 0) { *maxSVPtr = 0; return 0; }
    for (i = 0; i < n; i++) count[src[i]]++;
    while (!count[maxSV]) maxSV--;
    *maxSVPtr = maxSV;
    for (s = 0; s <= maxSV; s++) if (count[s] > largest) largest = count[s];
    return largest;
}

/* ------------------------------------------------------------------ FSE */
/* fse_compress.c:348-355 */
static u32 zo_fse_minTableLog(size_t srcSize, u32 maxSV)
{
    u32 minBitsSrc = zo_highbit32((u32)srcSize) + 1;
    u32 minBitsSymbols = zo_highbit32(maxSV) + 2;
    return minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols;
}
/* fse_compress.c:357-369 FSE_optimalTableLog_internal */
u32 zo_fse_optimalTableLog(u32 maxTableLog, size_t srcSize, u32 maxSV, u32 minus)
{
    u32 maxBitsSrc = zo_highbit32((u32)(srcSize - 1)) - minus;
    u32 tableLog = maxTableLog;
    u32 minBits = zo_fse_minTableLog(srcSize, maxSV);
    if (tableLog == 0) tableLog = 11;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < ZO_FSE_MIN_TABLELOG) tableLog = ZO_FSE_MIN_TABLELOG;
    if (tableLog > ZO_FSE_MAX_TABLELOG) tableLog = ZO_FSE_MAX_TABLELOG;
    return tableLog;
}

/* fse_compress.c:379-463 FSE_normalizeM2 (secondary normalisation) */
static size_t zo_fse_normalizeM2(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSV, s16 lowProbCount)
{
    const s16 NOT_YET = -2;
    u32 s, distributed = 0, toDistribute;
    u32 const lowThreshold = (u32)(total >> tableLog);
    u32 lowOne = (u32)((total * 3) >> (tableLog + 1));

    for (s = 0; s <= maxSV; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NOT_YET;
    }
    toDistribute = (1u << tableLog) - distributed;
    if (toDistribute == 0) return 0;

    if ((total / toDistribute) > lowOne) {
        lowOne = (u32)((total * 3) / (toDistribute * 2));
        for (s = 0; s <= maxSV; s++) {
            if (norm[s] == NOT_YET && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        }
        toDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSV + 1) {
        u32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSV; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (s16)toDistribute;
        return 0;
    }
    if (total == 0) {
        for (s = 0; toDistribute > 0; s = (s + 1) % (maxSV + 1))
            if (norm[s] > 0) { toDistribute--; norm[s]++; }
        return 0;
    }
    {   u64 const vStepLog = 62 - tableLog;
        u64 const mid = ((u64)1 << (vStepLog - 1)) - 1;
        u64 const rStep = ((((u64)1 << vStepLog) * toDistribute) + mid) / (u32)total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSV; s++) {
            if (norm[s] == NOT_YET) {
                u64 const end = tmpTotal + (count[s] * rStep);
                u32 const sStart = (u32)(tmpTotal >> vStepLog);
                u32 const sEnd = (u32)(end >> vStepLog);
                u32 const weight = sEnd - sStart;
                if (weight < 1) return ZO_ERR(GENERIC);
                norm[s] = (s16)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

/* fse_compress.c:465-525 FSE_normalizeCount */
size_t zo_fse_normalizeCount(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSV, u32 useLowProb)
{
    static const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    if (tableLog == 0) tableLog = 11;
    if (tableLog < ZO_FSE_MIN_TABLELOG) return ZO_ERR(GENERIC);
    if (tableLog > ZO_FSE_MAX_TABLELOG) return ZO_ERR(tableLog_tooLarge);
    if (tableLog < zo_fse_minTableLog(total, maxSV)) return ZO_ERR(GENERIC);
    {   s16 const lowProbCount = useLowProb ? -1 : 1;
        u64 const scale = 62 - tableLog;
        u64 const step = ((u64)1 << 62) / (u32)total;
        u64 const vStep = (u64)1 << (scale - 20);
        int stillToDistribute = 1 << tableLog;
        u32 s, largest = 0;
        s16 largestP = 0;
        u32 const lowThreshold = (u32)(total >> tableLog);

        for (s = 0; s <= maxSV; s++) {
            if (count[s] == total) return 0;   /* rle special case */
            if (count[s] == 0) { norm[s] = 0; continue; }
            if (count[s] <= lowThreshold) {
                norm[s] = lowProbCount;
                stillToDistribute--;
            } else {
                s16 proba = (s16)((count[s] * step) >> scale);
                if (proba < 8) {
                    u64 const restToBeat = vStep * rtb[proba];
                    proba += (count[s] * step) - ((u64)proba << scale) > restToBeat;
                }
                if (proba > largestP) { largestP = proba; largest = s; }
                norm[s] = proba;
                stillToDistribute -= proba;
            }
        }
        if (-stillToDistribute >= (norm[largest] >> 1)) {
            size_t const e = zo_fse_normalizeM2(norm, tableLog, count, total, maxSV, lowProbCount);
            if (zo_isErr(e)) return e;
        } else norm[largest] += (s16)stillToDistribute;
    }
    return tableLog;
}

/* fse_compress.c:234-340 FSE_writeNCount (bounds-checked form) */
size_t zo_fse_writeNCount(u8* dst, size_t cap, const s16* norm, u32 maxSV, u32 tableLog)
{
    u8* out = dst;
    u8* const oend = dst + cap;
    int nbBits, remaining, threshold, bitCount = 0, previousIs0 = 0;
    int const tableSize = 1 << tableLog;
    u32 bitStream = 0, symbol = 0;
    u32 const alphabetSize = maxSV + 1;

    if (tableLog > ZO_FSE_MAX_TABLELOG) return ZO_ERR(tableLog_tooLarge);
    if (tableLog < ZO_FSE_MIN_TABLELOG) return ZO_ERR(GENERIC);

    bitStream += (tableLog - ZO_FSE_MIN_TABLELOG) << bitCount;
    bitCount += 4;
    remaining = tableSize + 1;
    threshold = tableSize;
    nbBits = (int)tableLog + 1;

    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            u32 start = symbol;
            while ((symbol < alphabetSize) && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24;
                bitStream += 0xFFFFU << bitCount;
                if (out > oend - 2) return ZO_ERR(dstSize_tooSmall);
                out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2;
                bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3u << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount;
            bitCount += 2;
            if (bitCount > 16) {
                if (out > oend - 2) return ZO_ERR(dstSize_tooSmall);
                out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2;
                bitStream >>= 16; bitCount -= 16;
            }
        }
        {   int count = norm[symbol++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (u32)count << bitCount;
            bitCount += nbBits;
            bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return ZO_ERR(GENERIC);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) {
            if (out > oend - 2) return ZO_ERR(dstSize_tooSmall);
            out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2;
            bitStream >>= 16; bitCount -= 16;
        }
    }
    if (remaining != 1) return ZO_ERR(GENERIC);
    if (out > oend - 2) return ZO_ERR(dstSize_tooSmall);
    out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (size_t)(out - dst);
}

/* fse_compress.c:69-215 FSE_buildCTable_wksp.  The two spreading code paths there (:117-168) place the
 * same symbols in the same cells; the generic one is restated. */
void zo_fse_buildCTable(u16* nextState, int* dfs, u32* dnb, const s16* norm, u32 maxSV, u32 tableLog)
{
    u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1, step = ZO_FSE_TABLESTEP(tableSize);
    u32 const maxSV1 = maxSV + 1;
    u16 cumul[256 + 2];
    u8  tableSymbol[1 << ZO_FSE_MAX_TABLELOG];
    u32 highThreshold = tableSize - 1, u, s;

    cumul[0] = 0;
    for (u = 1; u <= maxSV1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[highThreshold--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
    }
    cumul[maxSV1] = (u16)(tableSize + 1);
    {   u32 position = 0;
        for (s = 0; s < maxSV1; s++) {
            int n, freq = norm[s];
            for (n = 0; n < freq; n++) {
                tableSymbol[position] = (u8)s;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    for (u = 0; u < tableSize; u++) { u8 sy = tableSymbol[u]; nextState[cumul[sy]++] = (u16)(tableSize + u); }
    {   u32 total = 0;
        for (s = 0; s <= maxSV; s++) {
            switch (norm[s]) {
            case 0:  dnb[s] = ((tableLog + 1) << 16) - (1u << tableLog); dfs[s] = 0; break;
            case -1:
            case 1:  dnb[s] = (tableLog << 16) - (1u << tableLog); dfs[s] = (int)(total - 1); total++; break;
            default: {
                u32 const maxBitsOut = tableLog - zo_highbit32((u32)norm[s] - 1);
                u32 const minStatePlus = (u32)norm[s] << maxBitsOut;
                dnb[s] = (maxBitsOut << 16) - minStatePlus;
                dfs[s] = (int)(total - (u32)norm[s]);
                total += (u32)norm[s];
            } }
        }
    }
}

/* fse_compress.c:528-548 FSE_buildCTable_rle */
static void zo_seq_ctable_rle(zo_seq_ctable* ct, u8 symbol)
{
    ct->tableLog = 0; ct->maxSV = symbol;
    ct->nextState[0] = 0; ct->nextState[1] = 0;
    ct->deltaNbBits[symbol] = 0; ct->deltaFindState[symbol] = 0;
}

/* FSE encoder state (fse.h:437-476) */
typedef struct { u32 value; const u16* next; const int* dfs; const u32* dnb; u32 stateLog; } zo_fse_cstate;
static void zo_fse_initCState2(zo_fse_cstate* st, const u16* next, const int* dfs, const u32* dnb, u32 tableLog, u32 symbol)
{
    u32 const nbBitsOut = (dnb[symbol] + (1u << 15)) >> 16;
    st->next = next; st->dfs = dfs; st->dnb = dnb; st->stateLog = tableLog;
    st->value = (nbBitsOut << 16) - dnb[symbol];
    st->value = next[(int)(st->value >> nbBitsOut) + dfs[symbol]];
}
static void zo_fse_encodeSymbol(zo_bitw* w, zo_fse_cstate* st, u32 symbol)
{
    u32 const nbBitsOut = (st->value + st->dnb[symbol]) >> 16;
    zo_bw_add(w, st->value, nbBitsOut);
    st->value = st->next[(int)(st->value >> nbBitsOut) + st->dfs[symbol]];
}
static void zo_fse_flushCState(zo_bitw* w, const zo_fse_cstate* st) { zo_bw_add(w, st->value, st->stateLog); }

/* ------------------------------------------------------------------ Huffman */
typedef struct { u32 count; u16 parent; u8 byte; u8 nbBits; } zo_node;

/* huf_compress.c:506-510 */
static u32 zo_huf_bucket(u32 count) { return (count < 166) ? count : zo_highbit32(count) + 158; }

/* huf_compress.c:531-583: insertion sort below 8 elements, else quicksort with last-element pivot */
static void zo_huf_insertionSort(zo_node* a, int low, int high)
{
    int i, size = high - low + 1;
    a += low;
    for (i = 1; i < size; ++i) {
        zo_node key = a[i];
        int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
static int zo_huf_partition(zo_node* a, int low, int high)
{
    u32 const pivot = a[high].count;
    int i = low - 1, j;
    for (j = low; j < high; j++) {
        if (a[j].count > pivot) { zo_node t; i++; t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    { zo_node t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
    return i + 1;
}
static void zo_huf_quickSort(zo_node* a, int low, int high)
{
    if (high - low < 8) { zo_huf_insertionSort(a, low, high); return; }
    while (low < high) {
        int const idx = zo_huf_partition(a, low, high);
        if (idx - low < high - idx) { zo_huf_quickSort(a, low, idx - 1); low = idx + 1; }
        else { zo_huf_quickSort(a, idx + 1, high); high = idx - 1; }
    }
}
/* huf_compress.c:596-641 HUF_sort */
static void zo_huf_sort(zo_node* node, const u32* count, u32 maxSV)
{
    struct { u16 base, curr; } rp[192];
    u32 n, maxSV1 = maxSV + 1;
    memset(rp, 0, sizeof(rp));
    for (n = 0; n < maxSV1; ++n) rp[zo_huf_bucket(count[n])].base++;
    for (n = 191; n > 0; --n) { rp[n - 1].base += rp[n].base; rp[n - 1].curr = rp[n - 1].base; }
    for (n = 0; n < maxSV1; ++n) {
        u32 const c = count[n], r = zo_huf_bucket(c) + 1, pos = rp[r].curr++;
        node[pos].count = c; node[pos].byte = (u8)n;
    }
    for (n = 166; n < 191; ++n) {
        int const bucketSize = rp[n].curr - rp[n].base;
        if (bucketSize > 1) zo_huf_quickSort(node + rp[n].base, 0, bucketSize - 1);
    }
}
/* huf_compress.c:657-694 HUF_buildTree.  node0 = node - 1 holds the sentinel. */
#define ZO_STARTNODE 256
static int zo_huf_buildTree(zo_node* node, u32 maxSV)
{
    zo_node* const node0 = node - 1;
    int nonNullRank = (int)maxSV, lowS, lowN, nodeNb = ZO_STARTNODE, n, nodeRoot;
    while (node[nonNullRank].count == 0) nonNullRank--;
    lowS = nonNullRank; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) node[n].count = (u32)(1U << 30);
    node0[0].count = (u32)(1U << 31);
    while (nodeNb <= nodeRoot) {
        int const n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        int const n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (u16)nodeNb;
        nodeNb++;
    }
    node[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= ZO_STARTNODE; n--) node[n].nbBits = node[node[n].parent].nbBits + 1;
    for (n = 0; n <= nonNullRank; n++) node[n].nbBits = node[node[n].parent].nbBits + 1;
    return nonNullRank;
}
/* huf_compress.c:352-474 HUF_setMaxHeight */
static u32 zo_huf_setMaxHeight(zo_node* node, u32 lastNonNull, u32 targetNbBits)
{
    u32 const largestBits = node[lastNonNull].nbBits;
    if (largestBits <= targetNbBits) return largestBits;
    {   int totalCost = 0;
        u32 const baseCost = 1u << (largestBits - targetNbBits);
        int n = (int)lastNonNull;
        while (node[n].nbBits > targetNbBits) {
            totalCost += (int)(baseCost - (1u << (largestBits - node[n].nbBits)));
            node[n].nbBits = (u8)targetNbBits;
            n--;
        }
        while (node[n].nbBits == targetNbBits) --n;
        totalCost >>= (largestBits - targetNbBits);
        {   u32 const noSymbol = 0xF0F0F0F0;
            u32 rankLast[ZO_HUF_TABLELOG_MAX + 2];
            memset(rankLast, 0xF0, sizeof(rankLast));
            {   u32 currentNbBits = targetNbBits;
                int pos;
                for (pos = n; pos >= 0; pos--) {
                    if (node[pos].nbBits >= currentNbBits) continue;
                    currentNbBits = node[pos].nbBits;
                    rankLast[targetNbBits - currentNbBits] = (u32)pos;
                }
            }
            while (totalCost > 0) {
                u32 nBitsToDecrease = zo_highbit32((u32)totalCost) + 1;
                for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                    u32 const highPos = rankLast[nBitsToDecrease];
                    u32 const lowPos = rankLast[nBitsToDecrease - 1];
                    if (highPos == noSymbol) continue;
                    if (lowPos == noSymbol) break;
                    {   u32 const highTotal = node[highPos].count;
                        u32 const lowTotal = 2 * node[lowPos].count;
                        if (highTotal <= lowTotal) break;
                    }
                }
                while ((nBitsToDecrease <= ZO_HUF_TABLELOG_MAX) && (rankLast[nBitsToDecrease] == noSymbol))
                    nBitsToDecrease++;
                totalCost -= 1 << (nBitsToDecrease - 1);
                node[rankLast[nBitsToDecrease]].nbBits++;
                if (rankLast[nBitsToDecrease - 1] == noSymbol)
                    rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                if (rankLast[nBitsToDecrease] == 0)
                    rankLast[nBitsToDecrease] = noSymbol;
                else {
                    rankLast[nBitsToDecrease]--;
                    if (node[rankLast[nBitsToDecrease]].nbBits != targetNbBits - nBitsToDecrease)
                        rankLast[nBitsToDecrease] = noSymbol;
                }
            }
            while (totalCost < 0) {
                if (rankLast[1] == noSymbol) {
                    while (node[n].nbBits == targetNbBits) n--;
                    node[n + 1].nbBits--;
                    rankLast[1] = (u32)(n + 1);
                    totalCost++;
                    continue;
                }
                node[rankLast[1] + 1].nbBits--;
                rankLast[1]++;
                totalCost++;
            }
        }
    }
    return targetNbBits;
}
/* huf_compress.c:730-766 HUF_buildCTable_wksp (+ :706-728 HUF_buildCTableFromTree) */
static size_t zo_huf_buildCTable(zo_huf_ctable* ct, const u32* count, u32 maxSV, u32 maxNbBits)
{
    zo_node tbl[2 * 256 + 1];
    zo_node* const node = tbl + 1;
    int nonNullRank, n;
    u16 nbPerRank[ZO_HUF_TABLELOG_MAX + 1] = {0}, valPerRank[ZO_HUF_TABLELOG_MAX + 1] = {0};
    if (maxNbBits == 0) maxNbBits = 11;
    memset(tbl, 0, sizeof(tbl));
    zo_huf_sort(node, count, maxSV);
    nonNullRank = zo_huf_buildTree(node, maxSV);
    maxNbBits = zo_huf_setMaxHeight(node, (u32)nonNullRank, maxNbBits);
    if (maxNbBits > ZO_HUF_TABLELOG_MAX) return ZO_ERR(GENERIC);
    for (n = 0; n <= nonNullRank; n++) nbPerRank[node[n].nbBits]++;
    {   u16 min = 0;
        for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; }
    }
    memset(ct, 0, sizeof(*ct));   /* :1376-1381 unused symbols are zeroed so the table can be validated later */
    for (n = 0; n <= (int)maxSV; n++) ct->nbBits[node[n].byte] = node[n].nbBits;
    for (n = 0; n <= (int)maxSV; n++) { u16 v = valPerRank[ct->nbBits[n]]++; ct->val[n] = ct->nbBits[n] ? v : 0; }
    ct->tableLog = maxNbBits;
    return maxNbBits;
}
/* huf_compress.c:768-787 */
static size_t zo_huf_estimate(const zo_huf_ctable* ct, const u32* count, u32 maxSV)
{
    size_t nbBits = 0; u32 s;
    for (s = 0; s <= maxSV; ++s) nbBits += (size_t)ct->nbBits[s] * count[s];
    return nbBits >> 3;
}
static int zo_huf_validate(const zo_huf_ctable* ct, const u32* count, u32 maxSV)
{
    int bad = 0; u32 s;
    for (s = 0; s <= maxSV; ++s) bad |= (count[s] != 0) & (ct->nbBits[s] == 0);
    return !bad;
}

/* fse_compress.c:551-608 FSE_compress_usingCTable_generic, as used on Huffman weights */
static size_t zo_fse_compressWeightsStream(u8* dst, const u8* src, size_t srcSize, const zo_fse_ctable* ct)
{
    const u8* ip = src + srcSize;
    zo_bitw w;
    zo_fse_cstate s1, s2;
    if (srcSize <= 2) return 0;
    zo_bw_init(&w, dst, NULL);
    if (srcSize & 1) {
        zo_fse_initCState2(&s1, ct->nextState, ct->deltaFindState, ct->deltaNbBits, ct->tableLog, *--ip);
        zo_fse_initCState2(&s2, ct->nextState, ct->deltaFindState, ct->deltaNbBits, ct->tableLog, *--ip);
        zo_fse_encodeSymbol(&w, &s1, *--ip);
    } else {
        zo_fse_initCState2(&s2, ct->nextState, ct->deltaFindState, ct->deltaNbBits, ct->tableLog, *--ip);
        zo_fse_initCState2(&s1, ct->nextState, ct->deltaFindState, ct->deltaNbBits, ct->tableLog, *--ip);
    }
    srcSize -= 2;
    if (srcSize & 2) { zo_fse_encodeSymbol(&w, &s2, *--ip); zo_fse_encodeSymbol(&w, &s1, *--ip); }
    while (ip > src) {
        zo_fse_encodeSymbol(&w, &s2, *--ip);
        zo_fse_encodeSymbol(&w, &s1, *--ip);
        zo_fse_encodeSymbol(&w, &s2, *--ip);
        zo_fse_encodeSymbol(&w, &s1, *--ip);
    }
    zo_fse_flushCState(&w, &s2);
    zo_fse_flushCState(&w, &s1);
    return zo_bw_close(&w);
}
/* huf_compress.c:147-187 HUF_compressWeights */
static size_t zo_huf_compressWeights(u8* dst, size_t dstSize, const u8* weights, size_t wtSize)
{
    u8* op = dst;
    u32 maxSV = ZO_HUF_TABLELOG_MAX, tableLog = 6;
    u32 count[ZO_HUF_TABLELOG_MAX + 1];
    s16 norm[ZO_HUF_TABLELOG_MAX + 1];
    zo_fse_ctable ct;
    if (wtSize <= 1) return 0;
    {   u32 const maxCount = zo_hist(count, &maxSV, weights, wtSize);
        if (maxCount == wtSize) return 1;
        if (maxCount == 1) return 0;
    }
    tableLog = zo_fse_optimalTableLog(tableLog, wtSize, maxSV, 2);
    { size_t const e = zo_fse_normalizeCount(norm, tableLog, count, wtSize, maxSV, 0); if (zo_isErr(e)) return e; }
    { size_t const h = zo_fse_writeNCount(op, dstSize, norm, maxSV, tableLog); if (zo_isErr(h)) return h; op += h; }
    ct.tableLog = (u16)tableLog; ct.maxSV = (u16)maxSV;
    zo_fse_buildCTable(ct.nextState, ct.deltaFindState, ct.deltaNbBits, norm, maxSV, tableLog);
    {   size_t const c = zo_fse_compressWeightsStream(op, weights, wtSize, &ct);
        if (c == 0) return 0;
        op += c;
    }
    return (size_t)(op - dst);
}
/* huf_compress.c:230-268 HUF_writeCTable_wksp */
static size_t zo_huf_writeCTable(u8* dst, size_t maxDstSize, const zo_huf_ctable* ct, u32 maxSV, u32 huffLog)
{
    u8 bitsToWeight[ZO_HUF_TABLELOG_MAX + 1];
    u8 huffWeight[256];
    u32 n;
    bitsToWeight[0] = 0;
    for (n = 1; n < huffLog + 1; n++) bitsToWeight[n] = (u8)(huffLog + 1 - n);
    for (n = 0; n < maxSV; n++) huffWeight[n] = bitsToWeight[ct->nbBits[n]];
    if (maxDstSize < 1) return ZO_ERR(dstSize_tooSmall);
    {   size_t const hSize = zo_huf_compressWeights(dst + 1, maxDstSize - 1, huffWeight, maxSV);
        if (zo_isErr(hSize)) return hSize;
        if ((hSize > 1) & (hSize < maxSV / 2)) { dst[0] = (u8)hSize; return hSize + 1; }
    }
    if (maxSV > (256 - 128)) return ZO_ERR(GENERIC);
    if (((maxSV + 1) / 2) + 1 > maxDstSize) return ZO_ERR(dstSize_tooSmall);
    dst[0] = (u8)(128 + (maxSV - 1));
    huffWeight[maxSV] = 0;
    for (n = 0; n < maxSV; n += 2) dst[(n / 2) + 1] = (u8)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSV + 1) / 2) + 1;
}
/* huf_compress.c:1023-1086 + :941-950: one stream, symbols last->first, end mark */
/* end of the block's output buffer while a block is entropy-coded (set by zo_entropyCompressSeqStore): the Huffman streams and
 * the sequence bitstream stop there, as the reference's do (huf_compress.c:1040-1048, zstd_compress_sequences.c:303-306) */
static __thread u8* zo_block_oend;
static size_t zo_huf_encode1X(u8* dst, const u8* src, size_t srcSize, const zo_huf_ctable* ct)
{
    zo_bitw w;
    size_t n = srcSize;
    zo_bw_init(&w, dst, zo_block_oend);
    while (n > 0) { u8 const s = src[--n]; zo_bw_add(&w, ct->val[s], ct->nbBits[s]); }
    return zo_bw_close(&w);
}
/* huf_compress.c:1135-1183 HUF_compress4X_usingCTable_internal */
static size_t zo_huf_encode4X(u8* dst, const u8* src, size_t srcSize, const zo_huf_ctable* ct)
{
    size_t const segmentSize = (srcSize + 3) / 4;
    const u8* ip = src;
    u8* op = dst + 6;
    int i;
    if (srcSize < 12) return 0;
    for (i = 0; i < 3; i++) {
        size_t const c = zo_huf_encode1X(op, ip, segmentSize, ct);
        if (c == 0 || c > 65535) return 0;
        zo_wr16(dst + 2 * i, (u32)c);
        op += c; ip += segmentSize;
    }
    {   size_t const c = zo_huf_encode1X(op, ip, (size_t)(src + srcSize - ip), ct);
        if (c == 0 || c > 65535) return 0;
        op += c;
    }
    return (size_t)(op - dst);
}
/* huf_compress.c:1192-1207 HUF_compressCTable_internal; op is where the streams start (after the header) */
static size_t zo_huf_compressCTable(u8* ostart, u8* op, const u8* src, size_t srcSize, int fourStreams, const zo_huf_ctable* ct)
{
    size_t const c = fourStreams ? zo_huf_encode4X(op, src, srcSize, ct) : zo_huf_encode1X(op, src, srcSize, ct);
    if (c == 0) return 0;
    op += c;
    if ((size_t)(op - ostart) >= srcSize - 1) return 0;
    return (size_t)(op - ostart);
}
/* huf_compress.c:1298-1406 HUF_compress_internal (maxSymbolValue=255, huffLog=11, optimalDepth off at
 * strategy fast, zstd_compress_literals.c:164-168).  `oldTable` is nextHuf->CTable (in/out). */
static size_t zo_huf_compress(u8* dst, size_t dstSize, const u8* src, size_t srcSize, int fourStreams,
                              zo_huf_ctable* oldTable, int* repeat, int preferRepeat, int suspectUncompressible)
{
    u8* const ostart = dst;
    u8* op = ostart;
    u32 count[256];
    u32 maxSV = 255, huffLog = ZO_LitHufLog;
    zo_huf_ctable ct;

    if (!srcSize) return 0;
    if (!dstSize) return 0;
    if (srcSize > ZO_BLOCKSIZE_MAX) return ZO_ERR(srcSize_wrong);
    if (preferRepeat && *repeat == zo_rep_valid)
        return zo_huf_compressCTable(ostart, op, src, srcSize, fourStreams, oldTable);
    if (suspectUncompressible && srcSize >= (4096 * 10)) {
        size_t largestTotal = 0;
        { u32 m = 255; largestTotal += zo_hist(count, &m, src, 4096); }
        { u32 m = 255; largestTotal += zo_hist(count, &m, src + srcSize - 4096, 4096); }
        if (largestTotal <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {   u32 const largest = zo_hist(count, &maxSV, src, srcSize);
        if (largest == srcSize) { *ostart = src[0]; return 1; }
        if (largest <= (srcSize >> 7) + 4) return 0;
    }
    if (*repeat == zo_rep_check && !zo_huf_validate(oldTable, count, maxSV)) *repeat = zo_rep_none;
    if (preferRepeat && *repeat != zo_rep_none)
        return zo_huf_compressCTable(ostart, op, src, srcSize, fourStreams, oldTable);

    huffLog = zo_fse_optimalTableLog(huffLog, srcSize, maxSV, 1);   /* :1252-1255 */
    {   size_t const maxBits = zo_huf_buildCTable(&ct, count, maxSV, huffLog);
        if (zo_isErr(maxBits)) return maxBits;
        huffLog = (u32)maxBits;
    }
    {   size_t const hSize = zo_huf_writeCTable(op, dstSize, &ct, maxSV, huffLog);
        if (zo_isErr(hSize)) return hSize;
        if (*repeat != zo_rep_none) {
            size_t const oldSize = zo_huf_estimate(oldTable, count, maxSV);
            size_t const newSize = zo_huf_estimate(&ct, count, maxSV);
            if (oldSize <= hSize + newSize || hSize + 12 >= srcSize)
                return zo_huf_compressCTable(ostart, op, src, srcSize, fourStreams, oldTable);
        }
        if (hSize + 12ul >= srcSize) return 0;
        op += hSize;
        *repeat = zo_rep_none;
        *oldTable = ct;
    }
    return zo_huf_compressCTable(ostart, op, src, srcSize, fourStreams, &ct);
}

/* ------------------------------------------------------------------ literals section */
/* zstd_compress_literals.c:40-67 */
static size_t zo_noCompressLiterals(u8* dst, const u8* src, size_t srcSize)
{
    u32 const flSize = 1 + (srcSize > 31) + (srcSize > 4095);
    switch (flSize) {
    case 1: dst[0] = (u8)((u32)zo_set_basic + (srcSize << 3)); break;
    case 2: zo_wr16(dst, (u32)zo_set_basic + (1 << 2) + (u32)(srcSize << 4)); break;
    default: zo_wr24(dst, (u32)zo_set_basic + (3 << 2) + (u32)(srcSize << 4)); break;
    }
    memcpy(dst + flSize, src, srcSize);
    return srcSize + flSize;
}
/* zstd_compress_literals.c:82-108 */
static size_t zo_rleLiterals(u8* dst, const u8* src, size_t srcSize)
{
    u32 const flSize = 1 + (srcSize > 31) + (srcSize > 4095);
    switch (flSize) {
    case 1: dst[0] = (u8)((u32)zo_set_rle + (srcSize << 3)); break;
    case 2: zo_wr16(dst, (u32)zo_set_rle + (1 << 2) + (u32)(srcSize << 4)); break;
    default: zo_wr24(dst, (u32)zo_set_rle + (3 << 2) + (u32)(srcSize << 4)); break;
    }
    dst[flSize] = src[0];
    return flSize + 1;
}
static int zo_allBytesIdentical(const u8* src, size_t n) { size_t p; for (p = 1; p < n; p++) if (src[p] != src[0]) return 0; return 1; }

/* zstd_compress_literals.c:130-236 ZSTD_compressLiterals (strategies fast .. lazy2: same thresholds, :115-128) */
static size_t zo_compressLiterals(u8* dst, size_t dstCapacity, const u8* src, size_t srcSize,
                                  const zo_entropy* prev, zo_entropy* next,
                                  int disableLiteralCompression, int suspectUncompressible, int strategy)
{
    size_t const lhSize = 3 + (srcSize >= 1024) + (srcSize >= 16384);
    u32 singleStream = srcSize < 256;
    int hType = zo_set_compressed;
    size_t cLitSize;

    next->huf = prev->huf; next->huf_repeat = prev->huf_repeat;   /* :153 */
    if (disableLiteralCompression) return zo_noCompressLiterals(dst, src, srcSize);
    {   /* :115-128 ZSTD_minLiteralsToCompress, strategy 1: shift = min(9-1,3) = 3 -> 64 */
        size_t const mintc = (prev->huf_repeat == zo_rep_valid) ? 6 : (size_t)8 << 3;
        if (srcSize < mintc) return zo_noCompressLiterals(dst, src, srcSize);
    }
    {   int repeat = prev->huf_repeat;
        int const preferRepeat = (strategy < 4) && (srcSize <= 1024);   /* :166: strategy < ZSTD_lazy */
        if (repeat == zo_rep_valid && lhSize == 3) singleStream = 1;
        cLitSize = zo_huf_compress(dst + lhSize, dstCapacity - lhSize, src, srcSize, !singleStream,
                                   &next->huf, &repeat, preferRepeat, suspectUncompressible);
        if (repeat != zo_rep_none) hType = zo_set_repeat;
    }
    {   size_t const minGain = (srcSize >> 6) + 2;   /* zstd_compress_internal.h:597-603 */
        if ((cLitSize == 0) || (cLitSize >= srcSize - minGain) || zo_isErr(cLitSize)) {
            next->huf = prev->huf; next->huf_repeat = prev->huf_repeat;
            return zo_noCompressLiterals(dst, src, srcSize);
        }
    }
    if (cLitSize == 1) {
        if ((srcSize >= 8) || zo_allBytesIdentical(src, srcSize)) {
            next->huf = prev->huf; next->huf_repeat = prev->huf_repeat;
            return zo_rleLiterals(dst, src, srcSize);
        }
    }
    if (hType == zo_set_compressed) next->huf_repeat = zo_rep_check;
    switch (lhSize) {
    case 3: zo_wr24(dst, (u32)hType + ((u32)(!singleStream) << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 14)); break;
    case 4: zo_wr32(dst, (u32)hType + (2 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 18)); break;
    default: zo_wr32(dst, (u32)hType + (3 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 22));
             dst[4] = (u8)(cLitSize >> 10); break;
    }
    return lhSize + cLitSize;
}

/* ------------------------------------------------------------------ sequences section */
/* zstd_compress_internal.h:523-552 */
static u32 zo_LLcode(u32 ll)
{
    static const u8 t[64] = { 0,1,2,3,4,5,6,7, 8,9,10,11,12,13,14,15, 16,16,17,17,18,18,19,19, 20,20,20,20,21,21,21,21,
                              22,22,22,22,22,22,22,22, 23,23,23,23,23,23,23,23, 24,24,24,24,24,24,24,24, 24,24,24,24,24,24,24,24 };
    return (ll > 63) ? zo_highbit32(ll) + 19 : t[ll];
}
static u32 zo_MLcode(u32 mlBase)
{
    if (mlBase > 127) return zo_highbit32(mlBase) + 36;
    if (mlBase < 32) return mlBase;
    if (mlBase < 40) return 32 + ((mlBase - 32) >> 1);
    if (mlBase < 48) return 36 + ((mlBase - 40) >> 2);
    if (mlBase < 64) return 38 + ((mlBase - 48) >> 3);
    if (mlBase < 96) return 40 + ((mlBase - 64) >> 4);
    return 42;
}
/* zstd_compress.c:2637-2663 ZSTD_seqToCodes */
static void zo_seqToCodes(const zo_seqstore* ss, u8* llCode, u8* ofCode, u8* mlCode)
{
    u32 u;
    for (u = 0; u < ss->nbSeq; u++) {
        llCode[u] = (u8)zo_LLcode(ss->seq[u].litLength);
        ofCode[u] = (u8)zo_highbit32(ss->seq[u].offBase);
        mlCode[u] = (u8)zo_MLcode(ss->seq[u].mlBase);
    }
    if (ss->longLengthType == 1) llCode[ss->longLengthPos] = ZO_MaxLL;
    if (ss->longLengthType == 2) mlCode[ss->longLengthPos] = ZO_MaxML;
}
/* zstd_compress_sequences.c:22-45 kInverseProbabilityLog256: floor(-log2(x / 256) * 256) for x in [1, 256), 0 for
 * x == 0 (tests/test_oracle.py re-derives the 255 values from that formula). */
static const u16 zo_invProbLog256[256] = {
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
    23,   22,   20,   19,   17,   16,   14,   13,   11,   10,   8,    7,    5,    4,    2,    1,
};
#define ZO_COST_ERROR ((size_t)-1)   /* ERROR(GENERIC): larger than any real cost */
/* zstd_compress_sequences.c:71-79 ZSTD_NCountCost: size of the table description a "compressed" mode would write */
static size_t zo_NCountCost(const u32* count, u32 max, size_t nbSeq, u32 FSELog)
{
    u8 wksp[512];
    s16 norm[ZO_MaxSeq + 1];
    u32 const tableLog = zo_fse_optimalTableLog(FSELog, nbSeq, max, 2);
    { size_t const e = zo_fse_normalizeCount(norm, tableLog, count, nbSeq, max, nbSeq >= 2048); if (zo_isErr(e)) return e; }
    return zo_fse_writeNCount(wksp, sizeof(wksp), norm, max, tableLog);
}
/* :85-99 ZSTD_entropyCost */
static size_t zo_entropyCost(const u32* count, u32 max, size_t total)
{
    u32 cost = 0, s;
    for (s = 0; s <= max; ++s) {
        u32 norm = (u32)((256 * count[s]) / total);
        if (count[s] != 0 && norm == 0) norm = 1;
        cost += count[s] * zo_invProbLog256[norm];
    }
    return cost >> 8;
}
/* :105-133 ZSTD_fseBitCost with fse.h:494-509 FSE_bitCost (accuracyLog 8) */
static size_t zo_fseBitCost(const zo_seq_ctable* ct, const u32* count, u32 max)
{
    u32 const kAccuracyLog = 8;
    u32 const tableLog = ct->tableLog;
    size_t cost = 0;
    u32 s;
    if (ct->maxSV < max) return ZO_COST_ERROR;
    for (s = 0; s <= max; ++s) {
        u32 const badCost = (tableLog + 1) << kAccuracyLog;
        u32 const minNbBits = ct->deltaNbBits[s] >> 16;
        u32 const threshold = (minNbBits + 1) << 16;
        u32 const tableSize = 1u << tableLog;
        u32 const deltaFromThreshold = threshold - (ct->deltaNbBits[s] + tableSize);
        u32 const normalizedDeltaFromThreshold = (deltaFromThreshold << kAccuracyLog) >> tableLog;
        u32 const bitCost = ((minNbBits + 1) << kAccuracyLog) - normalizedDeltaFromThreshold;
        if (count[s] == 0) continue;
        if (bitCost >= badCost) return ZO_COST_ERROR;
        cost += (size_t)count[s] * bitCost;
    }
    return cost >> kAccuracyLog;
}
/* :140-155 ZSTD_crossEntropyCost */
static size_t zo_crossEntropyCost(const s16* norm, u32 accuracyLog, const u32* count, u32 max)
{
    u32 const shift = 8 - accuracyLog;
    size_t cost = 0;
    u32 s;
    for (s = 0; s <= max; ++s) {
        u32 const normAcc = (norm[s] != -1) ? (u32)norm[s] : 1;
        u32 const norm256 = normAcc << shift;
        cost += count[s] * zo_invProbLog256[norm256];
    }
    return cost >> 8;
}
/* zstd_compress_sequences.c:157-236 ZSTD_selectEncodingType: the threshold rules below ZSTD_lazy (:180-205), the cost
 * estimates from ZSTD_lazy on (:206-232) */
static int zo_selectEncodingType(int* repeatMode, const u32* count, u32 max, u32 mostFrequent, size_t nbSeq, u32 FSELog,
                                 const zo_seq_ctable* prevCTable, const s16* defaultNorm, u32 defaultNormLog,
                                 int isDefaultAllowed, int strategy)
{
    if (mostFrequent == nbSeq) {
        *repeatMode = zo_rep_none;
        if (isDefaultAllowed && nbSeq <= 2) return zo_set_basic;
        return zo_set_rle;
    }
    if (strategy < 4) {   /* < ZSTD_lazy */
        if (isDefaultAllowed) {
            size_t const staticFse_nbSeq_max = 1000;
            size_t const mult = (size_t)(10 - strategy);   /* :183 */
            size_t const dynamicFse_nbSeq_min = (((size_t)1 << defaultNormLog) * mult) >> 3;
            if ((*repeatMode == zo_rep_valid) && (nbSeq < staticFse_nbSeq_max)) return zo_set_repeat;
            if ((nbSeq < dynamicFse_nbSeq_min) || (mostFrequent < (nbSeq >> (defaultNormLog - 1)))) {
                *repeatMode = zo_rep_none;
                return zo_set_basic;
            }
        }
    } else {
        size_t const basicCost = isDefaultAllowed ? zo_crossEntropyCost(defaultNorm, defaultNormLog, count, max) : ZO_COST_ERROR;
        size_t const repeatCost = *repeatMode != zo_rep_none ? zo_fseBitCost(prevCTable, count, max) : ZO_COST_ERROR;
        size_t const NCountCost = zo_NCountCost(count, max, nbSeq, FSELog);
        size_t const compressedCost = (NCountCost << 3) + zo_entropyCost(count, max, nbSeq);
        if (basicCost <= repeatCost && basicCost <= compressedCost) { *repeatMode = zo_rep_none; return zo_set_basic; }
        if (repeatCost <= compressedCost) return zo_set_repeat;
    }
    *repeatMode = zo_rep_check;
    return zo_set_compressed;
}
/* zstd_compress_sequences.c:243-289 ZSTD_buildCTable */
static size_t zo_buildSeqCTable(u8* dst, size_t cap, zo_seq_ctable* next, u32 FSELog, int type,
                                u32* count, u32 max, const u8* codeTable, size_t nbSeq,
                                const s16* defaultNorm, u32 defaultNormLog, u32 defaultMax,
                                const zo_seq_ctable* prev)
{
    switch (type) {
    case zo_set_rle:
        zo_seq_ctable_rle(next, (u8)max);
        if (cap == 0) return ZO_ERR(dstSize_tooSmall);
        *dst = codeTable[0];
        return 1;
    case zo_set_repeat:
        *next = *prev;
        return 0;
    case zo_set_basic:
        next->tableLog = (u16)defaultNormLog; next->maxSV = (u16)defaultMax;
        zo_fse_buildCTable(next->nextState, next->deltaFindState, next->deltaNbBits, defaultNorm, defaultMax, defaultNormLog);
        return 0;
    default: {
        s16 norm[ZO_MaxSeq + 1];
        size_t nbSeq_1 = nbSeq;
        u32 const tableLog = zo_fse_optimalTableLog(FSELog, nbSeq, max, 2);
        if (count[codeTable[nbSeq - 1]] > 1) { count[codeTable[nbSeq - 1]]--; nbSeq_1--; }
        { size_t const e = zo_fse_normalizeCount(norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048); if (zo_isErr(e)) return e; }
        {   size_t const NCountSize = zo_fse_writeNCount(dst, cap, norm, max, tableLog);
            if (zo_isErr(NCountSize)) return NCountSize;
            next->tableLog = (u16)tableLog; next->maxSV = (u16)max;
            zo_fse_buildCTable(next->nextState, next->deltaFindState, next->deltaNbBits, norm, max, tableLog);
            return NCountSize;
        }
    } }
}
/* zstd_compress_sequences.c:291-383 ZSTD_encodeSequences_body (64-bit: longOffsets == 0) */
static size_t zo_encodeSequences(u8* dst, const zo_seq_ctable* ctML, const u8* mlCode,
                                 const zo_seq_ctable* ctOF, const u8* ofCode,
                                 const zo_seq_ctable* ctLL, const u8* llCode,
                                 const zo_seqdef* seq, size_t nbSeq)
{
    zo_bitw w;
    zo_fse_cstate sML, sOF, sLL;
    size_t n;
    zo_bw_init(&w, dst, zo_block_oend);
    zo_fse_initCState2(&sML, ctML->nextState, ctML->deltaFindState, ctML->deltaNbBits, ctML->tableLog, mlCode[nbSeq - 1]);
    zo_fse_initCState2(&sOF, ctOF->nextState, ctOF->deltaFindState, ctOF->deltaNbBits, ctOF->tableLog, ofCode[nbSeq - 1]);
    zo_fse_initCState2(&sLL, ctLL->nextState, ctLL->deltaFindState, ctLL->deltaNbBits, ctLL->tableLog, llCode[nbSeq - 1]);
    zo_bw_add(&w, seq[nbSeq - 1].litLength, zo_LL_bits[llCode[nbSeq - 1]]);
    zo_bw_add(&w, seq[nbSeq - 1].mlBase, zo_ML_bits[mlCode[nbSeq - 1]]);
    zo_bw_add(&w, seq[nbSeq - 1].offBase, ofCode[nbSeq - 1]);
    for (n = nbSeq - 2; n < nbSeq; n--) {
        u8 const ll = llCode[n], of = ofCode[n], ml = mlCode[n];
        zo_fse_encodeSymbol(&w, &sOF, of);
        zo_fse_encodeSymbol(&w, &sML, ml);
        zo_fse_encodeSymbol(&w, &sLL, ll);
        zo_bw_add(&w, seq[n].litLength, zo_LL_bits[ll]);
        zo_bw_add(&w, seq[n].mlBase, zo_ML_bits[ml]);
        zo_bw_add(&w, seq[n].offBase, of);
    }
    zo_fse_flushCState(&w, &sML);
    zo_fse_flushCState(&w, &sOF);
    zo_fse_flushCState(&w, &sLL);
    return zo_bw_close(&w);
}

/* zstd_compress.c:2831-2986 ZSTD_entropyCompressSeqStore(_internal) + :2706-2824 ZSTD_buildSequencesStatistics.
 * Returns the compressed block body size, 0 for "emit raw", or an error. */
size_t zo_entropyCompressSeqStore(const zo_seqstore* ss, const zo_entropy* prev, zo_entropy* next,
                                  int literalsDisabled, u8* dst, size_t dstCapacity, size_t srcSize,
                                  u8* codeWksp /* 3*maxNbSeq bytes */, int strategy /* 1 fast .. 5 lazy2 */)
{
    u8* const ostart = dst;
    u8* const oend = dst + dstCapacity;
    u8* op = ostart;
    size_t const nbSeq = ss->nbSeq;
    u8* const llCode = codeWksp;
    u8* const ofCode = codeWksp + ss->maxNbSeq;
    u8* const mlCode = codeWksp + 2 * ss->maxNbSeq;
    size_t lastCountSize = 0;
    u32 count[ZO_MaxSeq + 1];

    zo_block_oend = oend;
    {   size_t const litSize = ss->litSize;
        int const suspect = (nbSeq == 0) || (litSize / nbSeq >= 20);
        size_t const c = zo_compressLiterals(op, dstCapacity, ss->lit, litSize, prev, next, literalsDisabled, suspect, strategy);
        if (zo_isErr(c)) return c;
        op += c;
    }
    if ((oend - op) < 3 + 1) return ZO_ERR(dstSize_tooSmall);
    if (nbSeq < 128) *op++ = (u8)nbSeq;
    else if (nbSeq < ZO_LONGNBSEQ) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; op += 2; }
    else { op[0] = 0xFF; zo_wr16(op + 1, (u32)(nbSeq - ZO_LONGNBSEQ)); op += 3; }
    if (nbSeq == 0) {
        next->ll = prev->ll; next->of = prev->of; next->ml = prev->ml;
        next->ll_repeat = prev->ll_repeat; next->of_repeat = prev->of_repeat; next->ml_repeat = prev->ml_repeat;
        goto check;
    }
    {   u8* const seqHead = op++;
        int LLtype, OFtype, MLtype;
        zo_seqToCodes(ss, llCode, ofCode, mlCode);
        {   u32 max = ZO_MaxLL;
            u32 const mostFrequent = zo_hist(count, &max, llCode, nbSeq);
            next->ll_repeat = prev->ll_repeat;
            LLtype = zo_selectEncodingType(&next->ll_repeat, count, max, mostFrequent, nbSeq, ZO_LLFSELog, &prev->ll,
                                           zo_LL_defaultNorm, ZO_LL_DEFAULTNORMLOG, 1, strategy);
            {   size_t const cs = zo_buildSeqCTable(op, (size_t)(oend - op), &next->ll, ZO_LLFSELog, LLtype, count, max, llCode, nbSeq,
                                                    zo_LL_defaultNorm, ZO_LL_DEFAULTNORMLOG, ZO_MaxLL, &prev->ll);
                if (zo_isErr(cs)) return cs;
                if (LLtype == zo_set_compressed) lastCountSize = cs;
                op += cs;
            }
        }
        {   u32 max = ZO_MaxOff;
            u32 const mostFrequent = zo_hist(count, &max, ofCode, nbSeq);
            int const defaultAllowed = (max <= ZO_DefaultMaxOff);
            next->of_repeat = prev->of_repeat;
            OFtype = zo_selectEncodingType(&next->of_repeat, count, max, mostFrequent, nbSeq, ZO_OffFSELog, &prev->of,
                                           zo_OF_defaultNorm, ZO_OF_DEFAULTNORMLOG, defaultAllowed, strategy);
            {   size_t const cs = zo_buildSeqCTable(op, (size_t)(oend - op), &next->of, ZO_OffFSELog, OFtype, count, max, ofCode, nbSeq,
                                                    zo_OF_defaultNorm, ZO_OF_DEFAULTNORMLOG, ZO_DefaultMaxOff, &prev->of);
                if (zo_isErr(cs)) return cs;
                if (OFtype == zo_set_compressed) lastCountSize = cs;
                op += cs;
            }
        }
        {   u32 max = ZO_MaxML;
            u32 const mostFrequent = zo_hist(count, &max, mlCode, nbSeq);
            next->ml_repeat = prev->ml_repeat;
            MLtype = zo_selectEncodingType(&next->ml_repeat, count, max, mostFrequent, nbSeq, ZO_MLFSELog, &prev->ml,
                                           zo_ML_defaultNorm, ZO_ML_DEFAULTNORMLOG, 1, strategy);
            {   size_t const cs = zo_buildSeqCTable(op, (size_t)(oend - op), &next->ml, ZO_MLFSELog, MLtype, count, max, mlCode, nbSeq,
                                                    zo_ML_defaultNorm, ZO_ML_DEFAULTNORMLOG, ZO_MaxML, &prev->ml);
                if (zo_isErr(cs)) return cs;
                if (MLtype == zo_set_compressed) lastCountSize = cs;
                op += cs;
            }
        }
        *seqHead = (u8)((LLtype << 6) + (OFtype << 4) + (MLtype << 2));
    }
    {   size_t const bitstreamSize = zo_encodeSequences(op, &next->ml, mlCode, &next->of, ofCode, &next->ll, llCode, ss->seq, nbSeq);
        /* :2929-2932 + :2961-2966: a bitstream that does not fit (dstSize_tooSmall) makes the block a raw one when the block
         * itself fits the buffer, else it is the frame's error */
        if (bitstreamSize == 0) return srcSize <= dstCapacity ? 0 : ZO_ERR(dstSize_tooSmall);
        op += bitstreamSize;
        if (lastCountSize && (lastCountSize + bitstreamSize) < 4) return 0;   /* :2938-2944 */
    }
check:
    {   size_t const cSize = (size_t)(op - ostart);
        size_t const maxCSize = srcSize - ((srcSize >> 6) + 2);   /* :2977 ZSTD_minGain */
        if (cSize >= maxCSize) return 0;
        return cSize;
    }
}
