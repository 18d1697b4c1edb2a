/*
 * oracle/misc.c -- synthetic data generator + batch drivers.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   orc_random_generator follows T/snappy/RandomGenerator.java:25-74 driven by
 *   java.util.Random(301) (LCG: seed = (seed * 0x5DEECE66D + 0xB) mod 2^48,
 *   next(bits) = seed >>> (48 - bits), nextInt(256) = (256 * next(31)) >> 31).
 *   T/ = src/test/java/io/airlift/compress/v3/
 */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
#include <string.h>

static uint64_t jseed;
static void jrandom_init(int64_t seed) { jseed = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1); }
static int32_t jrandom_next(int bits)
{
    jseed = (jseed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
    return (int32_t)(int64_t)(jseed >> (48 - bits));
}
static int32_t jrandom_next_int_256(void) { return (int32_t)((256LL * (int64_t)jrandom_next(31)) >> 31); }

void orc_random_generator(double compression_ratio, uint8_t* out, int64_t len)
{
    /* data = new byte[1048576 + 100]; fragments of 100 bytes until 1048576; positions past that stay 0 */
    uint8_t raw[100];
    jrandom_init(301);
    memset(out, 0, (size_t)len);
    for (int64_t i = 0; i < 1048576 && i < len; i += 100) {
        int32_t n = (int32_t)(100 * compression_ratio);
        if (n < 1) n = 1;
        for (int32_t k = 0; k < n; k++) raw[k] = (uint8_t)jrandom_next_int_256();
        for (int32_t j = 0; j < 100 && i + j < len;) {
            int32_t chunk = n < 100 - j ? n : 100 - j;
            for (int32_t k = 0; k < chunk && i + j + k < len; k++) out[i + j + k] = raw[k];
            j += chunk;
        }
    }
}

int64_t orc_batch(int32_t op, const uint8_t* src_base, const int64_t* src_off, const int32_t* src_len,
                  uint8_t* dst_base, const int64_t* dst_off, const int32_t* dst_cap,
                  int32_t* out_len, int32_t* status, int64_t* err_off, int32_t n_blocks)
{
    int64_t total = 0;
    for (int32_t i = 0; i < n_blocks; i++) {
        const uint8_t* s = src_base + src_off[i];
        uint8_t* d = dst_base + dst_off[i];
        int64_t eo = 0;
        int64_t r;
        switch (op) {
            case ACHIP_OP_LZ4_DECOMPRESS: r = orc_lz4_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_LZ4_COMPRESS: r = orc_lz4_compress(s, src_len[i], d, dst_cap[i]); break;
            case ACHIP_OP_SNAPPY_DECOMPRESS: r = orc_snappy_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_SNAPPY_COMPRESS: r = orc_snappy_compress(s, src_len[i], d, dst_cap[i]); break;
            case ACHIP_OP_ZSTD_DECOMPRESS: r = orc_zstd_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_ZSTD_COMPRESS: r = orc_zstd_compress(s, src_len[i], d, dst_cap[i]); break;
            /* the containers, at the Hadoop streams' default buffer size: bench.py's CPU legs beside the container extras */
            case ACHIP_OP_LZ4FRAME_DECOMPRESS: r = orc_lz4frame_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_LZ4FRAME_COMPRESS: r = orc_lz4frame_compress(s, src_len[i], d, dst_cap[i]); break;
            case ACHIP_OP_SNAPPYFRAMED_DECOMPRESS: r = orc_snappyframed_decompress(s, src_len[i], d, dst_cap[i], &eo); break;
            case ACHIP_OP_SNAPPYFRAMED_COMPRESS: r = orc_snappyframed_compress(s, src_len[i], d, dst_cap[i]); break;
            case ACHIP_OP_LZ4HADOOP_DECOMPRESS: r = orc_hadoop_decompress(0, s, src_len[i], d, dst_cap[i], 262144, &eo); break;
            case ACHIP_OP_LZ4HADOOP_COMPRESS: r = orc_hadoop_compress(0, s, src_len[i], d, dst_cap[i], 262144); break;
            case ACHIP_OP_SNAPPYHADOOP_DECOMPRESS: r = orc_hadoop_decompress(1, s, src_len[i], d, dst_cap[i], 262144, &eo); break;
            case ACHIP_OP_SNAPPYHADOOP_COMPRESS: r = orc_hadoop_compress(1, s, src_len[i], d, dst_cap[i], 262144); break;
            case ACHIP_OP_ZSTDSTREAM_COMPRESS: r = orc_zstd_stream_compress(s, src_len[i], d, dst_cap[i]); break;
            default: r = ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT); break;
        }
        if (r >= 0) {
            if (out_len) out_len[i] = (int32_t)r;
            if (status) status[i] = 0;
            if (err_off) err_off[i] = 0;
            total += r;
        }
        else {
            if (out_len) out_len[i] = 0;
            if (status) status[i] = (int32_t)r;
            if (err_off) err_off[i] = eo;
        }
    }
    return total;
}

/* ---- multi-threaded timing driver for bench.py's cpu_baseline leg ---------------------------------------------
 * The reference scales on a CPU the only way it can: one codec instance per thread, threads on disjoint block
 * ranges (SURVEY 2.3 / 8d "CPU baseline beside it").  T pthreads, thread t owns blocks [t*n/T, (t+1)*n/T) and
 * its own output ranges, and repeats passes over its range until `seconds` have elapsed (no Python in the loop:
 * round 1's ThreadPoolExecutor version measured dispatch overhead).  Returns plaintext bytes per second summed
 * over the threads (decompress: bytes produced; compress: bytes consumed), *passes = mean passes per thread. */
#include <pthread.h>
#include <time.h>

typedef struct {
    int32_t op;
    const uint8_t* src_base;
    const int64_t* src_off;
    const int32_t* src_len;
    uint8_t* dst_base;
    const int64_t* dst_off;
    const int32_t* dst_cap;
    int32_t first, last;
    double seconds;
    pthread_barrier_t* start;
    int64_t plain_bytes;
    int64_t passes;
    int64_t failures;
    double elapsed;
} orc_bench_task;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* orc_bench_thread(void* arg)
{
    orc_bench_task* t = (orc_bench_task*)arg;
    const int compress = (t->op & 1) || t->op == ACHIP_OP_ZSTDSTREAM_COMPRESS;
    pthread_barrier_wait(t->start);
    const double t0 = now_s();
    double t1 = t0;
    do {
        for (int32_t i = t->first; i < t->last; i++) {
            int32_t ol = 0, st = 0;
            int64_t eo = 0;
            orc_batch(t->op, t->src_base, t->src_off + i, t->src_len + i, t->dst_base, t->dst_off + i, t->dst_cap + i, &ol, &st, &eo, 1);
            if (st != 0) t->failures++;
            t->plain_bytes += compress ? t->src_len[i] : ol;
        }
        t->passes++;
        t1 = now_s();
    } while (t1 - t0 < t->seconds && t->last > t->first);
    t->elapsed = t1 - t0;
    orc_zstd_enc_thread_free();
    orc_zstd_dec_thread_free();
    return 0;
}

double orc_bench(int32_t op, const uint8_t* src_base, const int64_t* src_off, const int32_t* src_len, uint8_t* dst_base, const int64_t* dst_off,
                 const int32_t* dst_cap, int32_t n_blocks, int32_t threads, double seconds, double* passes, int64_t* failures)
{
    if (n_blocks > 0 && (op == ACHIP_OP_ZSTD_COMPRESS)) {  /* one-time table setup on the calling thread (not inside the timed threads) */
        static uint8_t warm_in[64], warm_out[256];
        (void)orc_zstd_compress(warm_in, sizeof(warm_in), warm_out, sizeof(warm_out));
    }
    if (threads < 1) threads = 1;
    if (threads > n_blocks) threads = n_blocks > 0 ? n_blocks : 1;
    pthread_t tid[1024];
    static orc_bench_task task[1024];
    if (threads > 1024) threads = 1024;
    pthread_barrier_t start;
    pthread_barrier_init(&start, 0, (unsigned)threads);
    for (int32_t t = 0; t < threads; t++) {
        orc_bench_task k = {op, src_base, src_off, src_len, dst_base, dst_off, dst_cap, (int32_t)((int64_t)n_blocks * t / threads),
                            (int32_t)((int64_t)n_blocks * (t + 1) / threads), seconds, &start, 0, 0, 0, 0.0};
        task[t] = k;
        pthread_create(&tid[t], 0, orc_bench_thread, &task[t]);
    }
    double rate = 0.0, p = 0.0;
    int64_t f = 0;
    for (int32_t t = 0; t < threads; t++) {
        pthread_join(tid[t], 0);
        if (task[t].elapsed > 0) rate += (double)task[t].plain_bytes / task[t].elapsed;
        p += (double)task[t].passes;
        f += task[t].failures;
    }
    pthread_barrier_destroy(&start);
    if (passes) *passes = p / threads;
    if (failures) *failures = f;
    return rate;
}
