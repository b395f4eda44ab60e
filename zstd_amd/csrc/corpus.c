/* corpus.c -- fixed-seed synthetic corpora for tests and bench.py:
 *   kind 0: log-like   syslog-style lines "<ISO-8601 ts> <host> <proc>[pid]: <LEVEL> <template with 1-3 fields>",
 *           200 templates drawn Zipf(1.1), 64 hosts;
 *   kind 1: JSON-like  newline-delimited objects with 8 keys (ts, host, service, status, latency_ms, msg, trace_id, env).
 * Every frame is generated independently from (seed, frame index), so any slice of a corpus can be produced on
 * any rank without generating the rest; frames are cut at byte boundaries (the last line is truncated).
 * Plain C, host only; built into libzcorpus.so by __graft_entry__.build().
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include <math.h>

static inline uint64_t sm64(uint64_t* s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static inline uint32_t rnd(uint64_t* s, uint32_t n) { return (uint32_t)((sm64(s) >> 33) % n); }

static const char* const VOCAB[] = {
 "connection","request","response","timeout","retry","upstream","backend","cache","miss","hit","user","session","token","expired",
 "refreshed","database","query","slow","index","scan","commit","rollback","transaction","lock","acquired","released","queue","depth",
 "worker","started","stopped","heartbeat","missed","node","joined","left","cluster","leader","elected","replica","lagging","bytes",
 "sent","received","from","to","in","ms","failed","succeeded","with","status","code","error","warning","disk","usage","above",
 "threshold","memory","pressure","gc","pause","collected","objects","thread","pool","exhausted","rejected","task","scheduled",
 "completed","job","batch","processed","records","skipped","invalid","payload","schema","mismatch","version","upgrade","config",
 "reloaded","certificate","renewed","handshake","tls","client","server","closed","reset","by","peer","accepted","listening","on",
 "port","rate","limit","exceeded","for","tenant","shard","rebalanced","snapshot","written","restored","checkpoint","flushed",
 "compaction","finished","segment","merged","deleted","created","updated","key","not","found","permission","denied","authenticated",
 "via","oauth","ldap","password","attempt","blocked","ip","address","route","added","removed","latency","p99","throughput","drop" };
#define NVOCAB (sizeof(VOCAB) / sizeof(VOCAB[0]))
static const char* const PROCS[] = { "nginx","sshd","kernel","postgres","redis","kubelet","dockerd","cron","systemd","envoy","app-server","scheduler" };
static const char* const LEVELS[] = { "INFO","INFO","INFO","INFO","WARN","ERROR","DEBUG","INFO" };
static const char* const SERVICES[] = { "api","auth","db","cache" };
static const char* const MSGS[] = { "request completed","cache miss, fetching from origin","upstream timeout, retrying","user login ok" };
static const int STATUSES[] = { 200, 200, 200, 404, 500 };

#define NTEMPL 200
typedef struct { uint8_t nw; uint8_t w[12]; uint8_t field[12]; /* 0 none, 1 dec, 2 hex, 3 ip */ } templ_t;
static templ_t g_templ[NTEMPL];
static double g_cdf[NTEMPL];
static int g_ready = 0;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static void init_templates(void)
{
    pthread_mutex_lock(&g_mu);
    if (!g_ready) {
        uint64_t s = 0xC0FFEE1234ULL;
        double tot = 0;
        for (int t = 0; t < NTEMPL; t++) {
            templ_t* T = &g_templ[t];
            T->nw = (uint8_t)(4 + rnd(&s, 7));
            int nf = 1 + (int)rnd(&s, 3);
            for (int i = 0; i < T->nw; i++) { T->w[i] = (uint8_t)rnd(&s, NVOCAB); T->field[i] = 0; }
            for (int f = 0; f < nf; f++) T->field[1 + rnd(&s, T->nw - 1)] = (uint8_t)(1 + rnd(&s, 3));
            tot += 1.0 / pow((double)(t + 1), 1.1);
            g_cdf[t] = tot;
        }
        for (int t = 0; t < NTEMPL; t++) g_cdf[t] /= tot;
        g_ready = 1;
    }
    pthread_mutex_unlock(&g_mu);
}

typedef struct { uint8_t* p; uint8_t* end; } out_t;
static inline void put(out_t* o, const char* s) { while (*s && o->p < o->end) *o->p++ = (uint8_t)*s++; }
static inline void putc_(out_t* o, char c) { if (o->p < o->end) *o->p++ = (uint8_t)c; }
static void putu(out_t* o, uint64_t v, int minDigits)
{
    char b[24]; int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n < minDigits) b[n++] = '0';
    while (n) putc_(o, b[--n]);
}
static void puthex(out_t* o, uint64_t v, int digits)
{
    for (int i = digits - 1; i >= 0; i--) putc_(o, "0123456789abcdef"[(v >> (4 * i)) & 15]);
}

static void gen_log_frame(uint8_t* dst, size_t n, uint64_t seed, uint64_t frame)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + frame * 0xD1B54A32D192ED03ULL + 1;
    out_t o = { dst, dst + n };
    uint64_t t_ms = 1767225600000ULL + frame * 977ULL * 13;   /* 2026-01-01 + drift per frame */
    while (o.p < o.end) {
        t_ms += rnd(&s, 40);
        uint64_t sec = t_ms / 1000, ms = t_ms % 1000;
        uint64_t day = sec / 86400, rem = sec % 86400;
        uint64_t dd = 1 + (day - 20454) % 28;   /* days since epoch -> a day-of-month in January-like range */
        put(&o, "2026-01-"); putu(&o, dd, 2); putc_(&o, 'T');
        putu(&o, rem / 3600, 2); putc_(&o, ':'); putu(&o, (rem / 60) % 60, 2); putc_(&o, ':'); putu(&o, rem % 60, 2);
        putc_(&o, '.'); putu(&o, ms, 3); put(&o, "Z host-"); putu(&o, rnd(&s, 64), 2); putc_(&o, ' ');
        put(&o, PROCS[rnd(&s, 12)]); putc_(&o, '['); putu(&o, 100 + rnd(&s, 30000), 1); put(&o, "]: ");
        put(&o, LEVELS[rnd(&s, 8)]); putc_(&o, ' ');
        {   double u = (double)(sm64(&s) >> 11) / 9007199254740992.0;
            int lo = 0, hi = NTEMPL - 1;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (g_cdf[mid] < u) lo = mid + 1; else hi = mid; }
            const templ_t* T = &g_templ[lo];
            for (int i = 0; i < T->nw; i++) {
                if (i) putc_(&o, ' ');
                put(&o, VOCAB[T->w[i]]);
                if (T->field[i] == 1) { putc_(&o, '='); putu(&o, rnd(&s, 100000), 1); }
                else if (T->field[i] == 2) { put(&o, "=0x"); puthex(&o, sm64(&s), 8); }
                else if (T->field[i] == 3) { put(&o, "=10."); putu(&o, rnd(&s, 256), 1); putc_(&o, '.'); putu(&o, rnd(&s, 256), 1); putc_(&o, '.'); putu(&o, rnd(&s, 256), 1); }
            }
        }
        putc_(&o, '\n');
    }
}

static void gen_json_frame(uint8_t* dst, size_t n, uint64_t seed, uint64_t frame)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + frame * 0xD1B54A32D192ED03ULL + 7;
    out_t o = { dst, dst + n };
    uint64_t ts = 1767225600ULL + frame * 61;
    while (o.p < o.end) {
        ts += rnd(&s, 3);
        put(&o, "{\"ts\":"); putu(&o, ts, 1);
        put(&o, ",\"host\":\"web-"); putu(&o, rnd(&s, 40), 2);
        put(&o, "\",\"service\":\""); put(&o, SERVICES[rnd(&s, 4)]);
        put(&o, "\",\"status\":"); putu(&o, (uint64_t)STATUSES[rnd(&s, 5)], 1);
        put(&o, ",\"latency_ms\":"); putu(&o, rnd(&s, 500), 1); putc_(&o, '.'); putu(&o, rnd(&s, 100), 2);
        put(&o, ",\"msg\":\""); put(&o, MSGS[rnd(&s, 4)]);
        put(&o, "\",\"trace_id\":\""); puthex(&o, sm64(&s), 16);
        put(&o, "\",\"env\":\"prod\"}\n");
    }
}

typedef struct { uint8_t* dst; size_t frameSize; uint64_t first, count, seed; int kind; } job_t;
static void* worker(void* a)
{
    job_t* j = (job_t*)a;
    for (uint64_t i = 0; i < j->count; i++) {
        uint8_t* p = j->dst + i * j->frameSize;
        if (j->kind == 0) gen_log_frame(p, j->frameSize, j->seed, j->first + i);
        else gen_json_frame(p, j->frameSize, j->seed, j->first + i);
    }
    return NULL;
}

/* Fill dst with nFrames frames of frameSize bytes: frames [firstFrame, firstFrame + nFrames) of corpus (kind, seed). */
void zc_generate(uint8_t* dst, size_t frameSize, uint64_t firstFrame, uint64_t nFrames, uint64_t seed, int kind, int nThreads)
{
    init_templates();
    if (nThreads < 1) nThreads = 1;
    if (nThreads > 64) nThreads = 64;
    if ((uint64_t)nThreads > nFrames) nThreads = nFrames ? (int)nFrames : 1;
    pthread_t th[64]; job_t jobs[64];
    uint64_t per = nFrames / (uint64_t)nThreads, extra = nFrames % (uint64_t)nThreads, at = 0;
    for (int t = 0; t < nThreads; t++) {
        uint64_t cnt = per + ((uint64_t)t < extra ? 1 : 0);
        jobs[t] = (job_t){ dst + at * frameSize, frameSize, firstFrame + at, cnt, seed, kind };
        at += cnt;
        if (t == nThreads - 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < nThreads - 1; t++) pthread_join(th[t], NULL);
}
