// achip_seqexec2.h -- the wavefront-per-block sequence executor, second version (lz4_decompress_v7.hip).
//
// What the first executor (achip_seqexec.h) taught: its time was inversely proportional to the wavefronts in flight -- it was waiting,
// not computing.  A wavefront waits for its vector memory operations in order and by count, and the first executor still had loads in
// the composing half of its loop (literal runs beyond 16 bytes, match sources just below the LDS window, long matches): each of them
// waited for itself AND for the prefetches of the next batch issued a moment earlier -- several memory latencies per batch of 64
// records -- and the `cur = nxt` hand-over at the loop's end copied registers whose loads were still in flight (another full wait).
// Here the rule is: every global load of a batch is issued in `prepare`, one batch ahead, unconditionally (lanes that need nothing
// read an address of their own that is always valid), and `compose` only touches LDS.  That needs records whose parts fit one 16-byte
// load: the parser (lz4_parse2_kernel) cuts every sequence into PIECES of at most 16 literal + 16 match bytes, and gives the later
// pieces of a long match the largest multiple of its offset that stays inside the match's periodic source region, so that they do
// not depend on the piece before them.
//
//   window   the last WIN bytes of the block's output live in a circular LDS window (win[p % WIN] = output byte p; the first 16 bytes
//            once more behind its end, so that a 16-byte read never wraps).  A batch is at most CAP = 1024 output bytes.
//   prepare  64 records -> two wave scans (output and compressed positions), one 16-byte load per lane for the literals (clamped into
//            the stream; the bytes wanted are shifted into place later), one for a match source that will have left the window by the
//            time the batch is composed (such bytes were flushed to the output buffer at least two batches ago), and the dependency
//            mask of a match that reads this batch's own output (two binary searches by lane shuffles).
//   compose  literals into the window (exact byte counts), far matches likewise, then the near matches in rounds -- a match is ready
//            when no lane it depends on is still pending --, then the finished 16-byte pieces leave for the output buffer, 1 KiB per
//            store instruction.
// The two batch contexts alternate roles (the loop is unrolled twice): nothing in flight is ever copied.
// Cross-lane operations only in wave-uniform control flow, data between lanes only across wave_sync(): runs under tools/hostemu.
#pragma once
#include "achip_seqexec.h"

namespace achip {
namespace sx2 {

using sx::CHUNK_RECS;
using sx::CHUNK_SLOTS;
using sx::rec_lit;
using sx::rec_ml;
using sx::rec_off;
using sx::rec_skip;
using sx::wave_bcast;
using sx::wave_scan_incl;

constexpr int WIN_DEFAULT = 4096;
constexpr int CAP = 1024;  // output bytes of one batch at the default window (a record is at most 32: a batch always takes at least 32 records)
// a batch may hold a quarter of the window (the far-match rule below wants everything older than WIN - CAP bytes flushed two batches ago);
// at 8 KiB 64 pieces of 32 bytes always fit a batch
template <int WIN>
constexpr int cap_of() { return WIN / 4; }
static_assert(cap_of<WIN_DEFAULT>() == CAP, "the default window's batch");

struct Batch {
    int32_t lit, ml, off;
    int32_t dstLit;     // output position of the literal run (the match follows at dstLit + lit)
    int32_t litShift;   // the literal bytes start this far into litData (0 except within the stream's last 16 bytes)
    int32_t far;        // 1: the match source is in farData
    u32x4 litData, farData;
    unsigned long long dep;  // lanes whose (pending) match this lane's match reads
    int32_t k;          // (uniform) records in the batch
    int32_t total;      // (uniform) output bytes
};

// v >> 8 * s bytes (s in 0..15), zero filled
__device__ __forceinline__ u32x4 shr_bytes(u32x4 v, int32_t s)
{
    uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
    if (s >= 8) {
        lo = hi;
        hi = 0;
        s -= 8;
    }
    if (s > 0) {
        lo = (lo >> (8 * s)) | (hi << (64 - 8 * s));
        hi >>= 8 * s;
    }
    return u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
}

// the first `off` bytes of v repeated to fill 16 bytes (1 <= off <= 15): what a match that overlaps itself produces
__device__ __forceinline__ u32x4 expand_period(u32x4 v, int32_t off)
{
    uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
#pragma unroll 1
    for (int32_t p = off; p < 16; p += p) {  // the first p bytes are good: copy them behind themselves
        if (p >= 8) {
            const uint64_t keep = p == 8 ? 0 : (hi & ((1ull << (8 * (p - 8))) - 1));
            hi = keep | (lo << (8 * (p - 8)));  // (bytes beyond 16 fall off)
        }
        else {
            const uint64_t m = (1ull << (8 * p)) - 1;
            const uint64_t l = lo & m;
            // l | l << 8p as a 128-bit value
            hi = 2 * p > 8 ? (l >> (64 - 8 * p)) : 0;
            lo = l | (l << (8 * p));
        }
    }
    return u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
}

template <int WIN>
struct Window {
    static constexpr int MASK = WIN - 1;
    uint8_t* win;  // LDS, WIN + 16 bytes

    // 16 bytes at output position p (resident bytes only are meaningful)
    __device__ __forceinline__ u32x4 read16(int32_t p) const
    {
        const uint8_t* s = win + (p & MASK);
        uint64_t a, b;
        __builtin_memcpy(&a, s, 8);
        __builtin_memcpy(&b, s + 8, 8);
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    }
    // The first n (0..16) bytes of v to output position p -- exactly those.  (Measured: redirecting the stores that are not wanted to a
    // per-lane sink instead of branching around them makes the kernel three times slower -- an unaligned LDS store costs by the lanes that
    // take part, and here most lanes skip most of the four.)
    __device__ __forceinline__ void write(int32_t p, u32x4 v, int32_t n)
    {
        const int32_t a = p & MASK;
        uint8_t* d = win + a;
        const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
        if (n <= 0) {
            return;
        }
        if (a < 16 || a + 16 > WIN) {  // at the window's ends: byte by byte, the first 16 bytes also into their copy behind the end
#pragma unroll 1
            for (int32_t i = 0; i < n; i++) {
                const int32_t q = (a + i) & MASK;
                const uint8_t byte = (uint8_t)((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xFF);
                win[q] = byte;
                if (q < 16) {
                    win[WIN + q] = byte;
                }
            }
            return;
        }
        if (n >= 16) {
            __builtin_memcpy(d, &lo, 8);
            __builtin_memcpy(d + 8, &hi, 8);
            return;
        }
        if (n & 8) __builtin_memcpy(d, &lo, 8);
        const uint64_t x8 = (n & 8) ? hi : lo;
        const uint32_t x4lo = (uint32_t)x8;
        if (n & 4) __builtin_memcpy(d + (n & 8), &x4lo, 4);
        const uint32_t x4 = (n & 4) ? (uint32_t)(x8 >> 32) : (uint32_t)x8;
        const uint16_t x2lo = (uint16_t)x4;
        if (n & 2) __builtin_memcpy(d + (n & 12), &x2lo, 2);
        const uint32_t x2 = (n & 2) ? x4 >> 16 : x4;
        if (n & 1) d[n & 14] = (uint8_t)x2;
    }
};

// the records of a batch: lane i gets record slot + i (i < nb); when the batch ends its chunk, lane nb gets the link.  (The load is
// unconditional: slots beyond the chunk read the arena's next 4 KiB, which exist -- the arena is allocated with a chunk to spare.)
__device__ __forceinline__ uint64_t load_records(const uint64_t* __restrict__ arena, int32_t chunk, int32_t slot, int lane)
{
    return arena[(int64_t)chunk * CHUNK_SLOTS + slot + lane];
}

// One block whose compressed stream is shorter than 16 bytes (the clamped literal loads below need 16): lane 0 alone, byte by byte.
__device__ __forceinline__ void exec_block_serial(const uint8_t* __restrict__ in, uint8_t* out, const uint64_t* __restrict__ arena, int32_t chunk, int32_t count, int lane)
{
    if (lane != 0) {
        return;
    }
    int32_t srcPos = 0, outPos = 0, slot = 0;
    for (int32_t i = 0; i < count; i++) {
        const uint64_t r = arena[(int64_t)chunk * CHUNK_SLOTS + slot];
        srcPos += rec_skip(r);
        for (int32_t k = 0; k < rec_lit(r); k++) {
            out[outPos++] = in[srcPos++];
        }
        for (int32_t k = 0; k < rec_ml(r); k++) {
            out[outPos] = out[outPos - rec_off(r)];
            outPos++;
        }
        slot++;
        if (slot == CHUNK_RECS) {
            chunk = (int32_t)(uint32_t)arena[(int64_t)chunk * CHUNK_SLOTS + CHUNK_RECS];
            slot = 0;
        }
    }
}

// What `prepare` and `compose` share for one block: the window, the output buffer, the literal source, the composing cursor.
template <int WIN>
struct Exec {
    static constexpr int MASK = WIN - 1;
    Window<WIN> io;
    uint8_t* out;
    const uint8_t* lit;  // where literal bytes come from (the compressed stream for LZ4 / Snappy, the literal buffer for Zstd)
    int32_t lastLoad;    // the last position of `lit` a 16-byte load may start at (>= 0: the callers see to that)
    int lane;
    int32_t outPos, flushPos;  // (uniform) compose cursor; everything below flushPos is in the output buffer

    __device__ __forceinline__ void init(uint8_t* win, uint8_t* out_, const uint8_t* lit_, int32_t litLen, int lane_)
    {
        io.win = win;
        out = out_;
        lit = lit_;
        lastLoad = litLen - 16;
        lane = lane_;
        outPos = 0;
        flushPos = 0;
    }

    // The second half of `prepare`: b.lit / ml / off / dstLit / k / total are set (this lane's piece; zero lengths beyond k), pOut is the
    // batch's first output position, srcLit this lane's literal source position.  Issues the batch's two loads -- unconditionally -- and
    // builds the dependency mask.
    __device__ __forceinline__ void finish_prepare(Batch& b, int32_t srcLit, int32_t pOut)
    {
        const int32_t lit_ = b.lit, ml = b.ml, off = b.off;
        const int32_t pEnd = pOut + b.total;
        // the literal bytes: one 16-byte load, clamped into the source
        const int32_t at = srcLit < lastLoad ? srcLit : lastLoad;
        b.litShift = srcLit - at;
        b.litData = ld16(lit + at);
        // the match source: from the output buffer when it will have left the window by the time this batch is composed (then it was
        // flushed long ago: the window reaches back at least WIN - CAP bytes from the batch's start)
        const int32_t dstM = b.dstLit + lit_;
        const int32_t srcM = dstM - off;
        const bool isFar = ml > 0 && srcM < pEnd - WIN;
        b.far = isFar ? 1 : 0;
        b.farData = ld16(isFar ? out + srcM : lit);
        // a near match whose source reaches into this batch's own output waits for exactly the lanes a .. bnd-1 that produce it (outputs are
        // contiguous and ordered over the lanes: two binary searches by lane shuffles), as far as their matches are near ones themselves
        const bool isNear = ml > 0 && !isFar;
        const int32_t span = ml < off ? ml : off;  // source bytes that are not the match's own output
        const bool inBatch = isNear && srcM + span > pOut;
        const unsigned long long producers = __ballot(isNear);
        unsigned long long dep = 0;
        if (__ballot(inBatch) != 0) {  // (uniform)
            const int32_t myStart = b.dstLit, myEnd = dstM + ml;
            int32_t a = 0, bnd = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const int32_t e = __shfl(myEnd, a + step - 1);
                const int32_t st = __shfl(myStart, bnd + step - 1);
                a += e <= srcM ? step : 0;            // lanes that end at or before the source's start
                bnd += st < srcM + span ? step : 0;   // lanes that start before the source's end
            }
            bnd = bnd < lane ? bnd : lane;
            if (inBatch && a < bnd) {
                dep = producers & ((bnd >= 64 ? ~0ull : ((1ull << bnd) - 1)) & ~((1ull << a) - 1));
            }
        }
        b.dep = dep;
    }

    __device__ __forceinline__ void compose(const Batch& b)
    {
        // (Everything LDS here sits under a branch that only the lanes concerned take -- deliberately: an unaligned LDS access costs by
        // the lanes that take part.  Reading and writing with all lanes and selecting afterwards, flags in vector registers as in the
        // parser, measured 22.8 ms against 17.3.)
        const int32_t dstM = b.dstLit + b.lit;
        // ---- literal runs and the matches whose bytes came with the batch ----
        if (b.lit > 0) {
            io.write(b.dstLit, b.litShift != 0 ? shr_bytes(b.litData, b.litShift) : b.litData, b.lit);
        }
        bool pending = b.ml > 0;
        if (pending && b.far != 0) {
            io.write(dstM, b.off < b.ml ? expand_period(b.farData, b.off) : b.farData, b.ml);
            pending = false;
        }
        wave_sync();
        // ---- near matches, in rounds: the first pending lane is always ready ----
        for (;;) {  // (uniform)
            const unsigned long long pm = __ballot(pending);
            if (pm == 0) {
                break;
            }
            if (pending && (b.dep & pm) == 0) {
                const u32x4 v = io.read16(dstM - b.off);
                io.write(dstM, b.off < b.ml ? expand_period(v, b.off) : v, b.ml);
                pending = false;
            }
            wave_sync();
        }
        outPos += b.total;
        // ---- the finished 16-byte pieces leave (positions, not addresses, are 16-aligned) ----
        const int32_t wholeEnd = outPos & ~15;
        for (int32_t base = flushPos; base < wholeEnd; base += 1024) {  // (uniform; at most two rounds)
            const int32_t p = base + lane * 16;
            if (p < wholeEnd) {
                st16(out + p, *(const u32x4*)(io.win + (p & MASK)));
            }
        }
        flushPos = wholeEnd;
        wave_sync();  // (the next batch writes the window)
    }

    __device__ __forceinline__ void finish()
    {
        wave_sync();
        if (flushPos + lane < outPos) {  // the last bytes (fewer than 16)
            out[flushPos + lane] = io.win[(flushPos + lane) & MASK];
        }
    }
};

// Runs `count` records (pieces: at most 16 literal and 16 match bytes each) of one block, starting at slot 0 of chunk `chunk`.
// in / inLen: the block's compressed bytes (literal source); out: its output.  win: WIN + 16 bytes of LDS owned by this wavefront.
// The records were validated by the parser: every literal range lies inside the input, every match source inside the output produced
// so far, the total inside the block's capacity.
template <int WIN = WIN_DEFAULT>
__device__ __forceinline__ void exec_block(uint8_t* win, const uint8_t* __restrict__ in, int32_t inLen, uint8_t* out, const uint64_t* __restrict__ arena, int32_t chunk,
                                           int32_t count, int lane)
{
    if (inLen < 16) {  // (uniform)
        exec_block_serial(in, out, arena, chunk, count, lane);
        return;
    }
    Exec<WIN> X;
    X.init(win, out, in, inLen, lane);

    // cursor of `prepare`
    int32_t pChunk = chunk, pSlot = 0, pCount = count, pOut = 0, pSrc = 0;

    auto prepare = [&](uint64_t r, Batch& b) {
        int32_t nb = CHUNK_RECS - pSlot;
        nb = nb < 64 ? nb : 64;
        nb = nb < pCount ? nb : pCount;
        const bool haveLink = pSlot + nb == CHUNK_RECS && nb < 64;
        const int32_t linkChunk = haveLink ? (int32_t)(uint32_t)sx::shfl_u64(r, nb) : 0;  // (uniform)
        if (lane >= nb) {
            r = 0;
        }
        int32_t lit = rec_lit(r), ml = rec_ml(r);
        const int32_t off = rec_off(r), skip = rec_skip(r);
        const int32_t tot = lit + ml, adv = skip + lit;
        const int32_t oEnd = wave_scan_incl(tot, lane), sEnd = wave_scan_incl(adv, lane);
        int32_t k = (int32_t)__popcll(__ballot(lane < nb && oEnd <= cap_of<WIN>()));  // a prefix: oEnd is monotone (>= 1: a piece is <= 32 bytes)
        k = k < 1 ? 1 : k;
        if (lane >= k) {
            lit = 0;
            ml = 0;
        }
        b.k = k;
        b.total = wave_bcast(oEnd, k - 1);
        const int32_t sTotal = wave_bcast(sEnd, k - 1);
        b.lit = lit;
        b.ml = ml;
        b.off = off;
        b.dstLit = pOut + oEnd - tot;
        X.finish_prepare(b, pSrc + sEnd - (tot - rec_ml(r)), pOut);
        pOut += b.total;
        pSrc += sTotal;
        pSlot += k;
        pCount -= k;
        if (pSlot == CHUNK_RECS && pCount > 0) {
            // (a batch of exactly 64 records that ends its chunk had no free lane for the link: read it now)
            pChunk = haveLink ? linkChunk : (int32_t)(uint32_t)arena[(int64_t)pChunk * CHUNK_SLOTS + CHUNK_RECS];
            pSlot = 0;
        }
    };

    Batch A, B;
    uint64_t rNext = load_records(arena, pChunk, pSlot, lane);
    prepare(rNext, A);
    rNext = load_records(arena, pChunk, pSlot, lane);
    int32_t left = count;
    while (left > 0) {  // (uniform) two batches per trip: the contexts swap roles, nothing in flight is copied
        if (pCount > 0) {
            prepare(rNext, B);
            rNext = load_records(arena, pChunk, pSlot, lane);
        }
        left -= A.k;
        X.compose(A);
        if (left <= 0) {
            break;
        }
        if (pCount > 0) {
            prepare(rNext, A);
            rNext = load_records(arena, pChunk, pSlot, lane);
        }
        left -= B.k;
        X.compose(B);
    }
    X.finish();
}

// ---- records of any length: the executor cuts them into pieces itself (zstd_decompress_pipe.hip) ---------------------------------------
// The Zstd sequence stage produces one record per sequence, {literal length, match length, offset} of up to 128 KiB each, and the literals
// lie in one buffer in order.  Here a GROUP of 64 records is scanned once (output and literal positions, pieces per record), and every
// batch takes the next 64 pieces of the group: a lane finds its piece's record by a binary search over the scanned piece counts (lane
// shuffles), derives the piece -- at most 16 literal bytes and, behind a record's last literal bytes, at most 16 match bytes; later match
// pieces name the largest multiple of the offset inside the periodic source region, as the LZ4 / Snappy parsers do -- and from there a
// batch is what it is above.  The records are NOT trusted: a group with a record that runs outside the output capacity or the literal
// buffer, or whose match starts before the output's first byte (ZstdFrameDecompressor.java:491-496), stops the block with `bad` set;
// what was executed before it is valid, the caller sends the item to its fallback.
// An offset field at or above REP_SENTINEL names a repeat offset the producer could not know -- the Zstd sequence stage decodes the blocks
// of a multi-block frame side by side, and a block's first repeat-offset codes refer to the history the block BEFORE it leaves behind:
// REP_SENTINEL | k << 16 | m  stands for  max(rep[k] - m, 1), rep[] being the history at the block's start (ZstdFrameDecompressor.java:419-452:
// every "offset - 1" step clamps at 1).  Real offsets are at most 1 << 24 (the stage rejects larger ones).
constexpr int32_t REP_SENTINEL = 1 << 25;
__device__ __forceinline__ int32_t rep_resolve(int32_t v, int32_t rep0, int32_t rep1, int32_t rep2)
{
    if (v < REP_SENTINEL) {
        return v;
    }
    const int32_t k = (v >> 16) & 3;
    const int32_t r = (k == 0 ? rep0 : (k == 1 ? rep1 : rep2)) - (v & 0xFFFF);
    return r < 1 ? 1 : r;
}

struct RecordSource {  // one block's records for exec_records: `n` records at `rec`, then the literals left over as one last run
    const uint64_t* rec;
    int32_t n;
    int32_t rep0, rep1, rep2;  // the repeat-offset history at the block's start (only read where a record holds a sentinel)
    // the record layout of the Zstd sequence stage: bits 0..17 literal length, 18..35 match length, 36.. offset
    __device__ __forceinline__ static int32_t lit_of(uint64_t r) { return (int32_t)(r & 0x3FFFF); }
    __device__ __forceinline__ static int32_t ml_of(uint64_t r) { return (int32_t)((r >> 18) & 0x3FFFF); }
    __device__ __forceinline__ static int32_t off_of(uint64_t r) { return (int32_t)(r >> 36); }
};

template <int WIN = WIN_DEFAULT>
__device__ __forceinline__ int32_t exec_records(uint8_t* win, const RecordSource& S, const uint8_t* __restrict__ lit, int32_t litSize, uint8_t* out, int32_t outLimit, int lane,
                                                bool& badOut, int32_t startPos = 0, bool warm = false)
{
    // startPos > 0: the block continues an output whose first startPos bytes this wavefront produced by earlier calls with the same
    // window (the blocks of one Zstd frame): positions, offsets and the capacity are those of the whole output, the window still holds
    // its last bytes, and everything below startPos is in the output buffer already.
    badOut = false;
    // the clamped 16-byte literal loads need 16 readable bytes: a shorter literal buffer is copied to LDS behind the window first
    __shared__ __attribute__((aligned(16))) uint8_t shortLit[64 * 0 + 16];
    const uint8_t* litSrc = lit;
    int32_t litLen = litSize;
    if (litSize < 16) {  // (uniform)
        if (lane < 16) {
            shortLit[lane] = lane < litSize ? lit[lane] : 0;
        }
        wave_sync();
        litSrc = shortLit;
        litLen = 16;
    }
    Exec<WIN> X;
    X.init(win, out, litSrc, litLen, lane);
    X.outPos = startPos;
    X.flushPos = startPos & ~15;  // (the bytes between were written by the call before, and are written again with the first flush)
    if (warm && startPos > 0) {  // (uniform) the output's last bytes were produced by ANOTHER launch (a stream decoded a step at a time): the window is filled from the output buffer
        const int32_t from = startPos > WIN ? startPos - WIN : 0;
        for (int32_t p = from + lane; p < startPos; p += 64) {
            const int32_t q = p & (WIN - 1);
            const uint8_t byte = out[p];
            win[q] = byte;
            if (q < 16) {
                win[WIN + q] = byte;
            }
        }
        wave_sync();
    }

    // ---- the group under way (per lane: one record of it) ----
    int32_t gLit = 0, gMl = 0, gOff = 0;
    int32_t gOutStart = 0, gLitStart = 0;  // where the record's output / literals start
    int32_t gPieceEnd = 0;                 // pieces of the records up to and including this one (inclusive scan)
    int32_t gPieces = 0;                   // (uniform) pieces of the group
    int32_t cursor = 0;                    // (uniform) pieces of the group already handed out
    int32_t nextRec = 0;                   // (uniform) first record of the next group; S.n = the last literals; S.n + 1 = nothing left
    int32_t gOut = startPos, gSrc = 0;     // (uniform) output / literal position behind the group
    bool bad = false;                      // (uniform)

    auto load_group = [&](int32_t first) -> uint64_t {  // (unconditional where there are records: indices beyond them read record 0)
        const int32_t i = first + lane;
        return S.n > 0 ? S.rec[i < S.n ? i : 0] : 0ull;
    };
    // takes the next group: scans it; returns false when nothing is left (or the group is bad)
    auto next_group = [&](uint64_t r) -> bool {
        for (;;) {  // (uniform) groups without pieces are skipped
            if (nextRec > S.n || bad) {
                return false;
            }
            int32_t nb = S.n - nextRec;
            nb = nb < 64 ? nb : 64;
            int32_t lit_ = 0, ml = 0, off = 0;
            if (lane < nb) {
                lit_ = RecordSource::lit_of(r);
                ml = RecordSource::ml_of(r);
                off = rep_resolve(RecordSource::off_of(r), S.rep0, S.rep1, S.rep2);
            }
            if (nb == 0) {  // behind the last record: the literals left over (copyLastLiteral :518-525)
                lit_ = lane == 0 ? litSize - gSrc : 0;
                nb = 1;
                nextRec = S.n + 1;
            }
            else {
                nextRec += nb;
            }
            const int32_t tot = lit_ + ml;
            const int32_t oEnd = wave_scan_incl(tot, lane), sEnd = wave_scan_incl(lit_, lane);
            gLit = lit_;
            gMl = ml;
            gOff = off;
            gOutStart = gOut + oEnd - tot;
            gLitStart = gSrc + sEnd - lit_;
            // ZstdFrameDecompressor.java:491-496 (and a negative "rest" when the sequences used more literals than there are)
            const bool wrong = lit_ < 0 || (int64_t)gOutStart + tot > outLimit || gLitStart + lit_ > litSize || (ml > 0 && (off <= 0 || off > gOutStart + lit_));
            if (__ballot(wrong) != 0) {  // (uniform)
                bad = true;
                return false;
            }
            // pieces: the literal pieces of 16 that carry no match, the piece with the last literal bytes and the first 16 match bytes,
            // the remaining match pieces
            const int32_t litFull = lit_ > 16 ? (lit_ + 15) / 16 - 1 : 0;
            const int32_t matchRest = ml > 16 ? (ml - 16 + 15) / 16 : 0;
            const int32_t pieces = tot > 0 ? litFull + 1 + matchRest : 0;
            gPieceEnd = wave_scan_incl(pieces, lane);
            gPieces = wave_bcast(gPieceEnd, 63);
            gOut += wave_bcast(oEnd, 63);
            gSrc += wave_bcast(sEnd, 63);
            cursor = 0;
            if (gPieces > 0) {
                return true;
            }
            r = load_group(nextRec);  // (rare: a group of empty records -- a synchronous load)
        }
    };

    // the next batch: up to 64 pieces of the group (fewer where CAP output bytes are reached)
    auto prepare = [&](Batch& b) {
        const int32_t q = cursor + lane;
        const bool valid = q < gPieces;
        // the record of piece q: the first lane whose inclusive piece count exceeds q
        int32_t rIdx = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const int32_t e = __shfl(gPieceEnd, rIdx + step - 1);
            rIdx += e <= q ? step : 0;
        }
        rIdx = rIdx > 63 ? 63 : rIdx;
        const int32_t rLit = __shfl(gLit, rIdx), rMl = __shfl(gMl, rIdx), rOff = __shfl(gOff, rIdx);
        const int32_t rOut = __shfl(gOutStart, rIdx), rSrc = __shfl(gLitStart, rIdx);
        const int32_t rEnd = __shfl(gPieceEnd, rIdx);
        const int32_t litFull = rLit > 16 ? (rLit + 15) / 16 - 1 : 0;
        const int32_t matchRest = rMl > 16 ? (rMl - 16 + 15) / 16 : 0;
        const int32_t kk = q - (rEnd - (litFull + 1 + matchRest));  // piece index within the record
        int32_t pl, pm, o = rOff, dst, src;
        if (kk < litFull) {
            pl = 16;
            pm = 0;
            dst = rOut + 16 * kk;
            src = rSrc + 16 * kk;
        }
        else if (kk == litFull) {
            pl = rLit - 16 * litFull;
            pm = rMl < 16 ? rMl : 16;
            dst = rOut + 16 * litFull;
            src = rSrc + 16 * litFull;
        }
        else {
            const int32_t m = kk - litFull;  // >= 1
            pl = 0;
            pm = rMl - 16 * m < 16 ? rMl - 16 * m : 16;
            dst = rOut + rLit + 16 * m;
            src = rSrc + rLit;
            const uint32_t x = 16u * (uint32_t)m + (uint32_t)rOff;  // (< 2^29: lengths <= 2^18, offsets < 2^28)
            o = rOff > 0 ? (int32_t)((x / (uint32_t)rOff) * (uint32_t)rOff) : 0;  // the largest multiple of the offset inside the periodic source region
        }
        if (!valid) {
            pl = 0;
            pm = 0;
        }
        const int32_t pOut = wave_bcast(dst, 0);  // (lane 0 is always valid)
        const int32_t endRel = dst + pl + pm - pOut;
        int32_t k = (int32_t)__popcll(__ballot(valid && endRel <= cap_of<WIN>()));  // a prefix: the pieces are contiguous and ordered
        k = k < 1 ? 1 : k;
        if (lane >= k) {
            pl = 0;
            pm = 0;
        }
        b.k = k;
        b.total = wave_bcast(endRel, k - 1);
        b.lit = pl;
        b.ml = pm;
        b.off = o;
        b.dstLit = dst;
        X.finish_prepare(b, src, pOut);
        cursor += k;
    };

    Batch A, B;
    uint64_t rNext = load_group(0);
    bool haveA = next_group(rNext), haveB = false;
    rNext = load_group(nextRec);
    if (haveA) {
        prepare(A);
    }
    while (haveA) {  // (uniform) two batches per trip, as in exec_block
        // ---- B: the batch behind A ----
        haveB = true;
        if (cursor >= gPieces) {
            haveB = next_group(rNext);
            rNext = load_group(nextRec);
        }
        if (haveB) {
            prepare(B);
        }
        X.compose(A);
        if (!haveB) {
            break;
        }
        // ---- A: the batch behind B ----
        haveA = true;
        if (cursor >= gPieces) {
            haveA = next_group(rNext);
            rNext = load_group(nextRec);
        }
        if (haveA) {
            prepare(A);
        }
        X.compose(B);
    }
    X.finish();
    badOut = bad;
    return X.outPos;
}

}  // namespace sx2
}  // namespace achip
