// lz4_decompress_v7.hip -- batched LZ4 block decode for gfx950 in two passes: parse to records, then a wavefront per block executes them
// (achip_seqexec.h: records and arena; achip_seqexec2.h: the executor; DESIGN 4c: the design and the measurements).
//
// Same contract and the same Java-order checks as the other LZ4 decoders (M/lz4/Lz4RawDecompressor.java:35-198).  Every check of the
// Java loop depends on lengths, offsets and positions only -- never on a decoded byte -- so the PARSE pass alone decides status, error
// offset and output length of a block exactly as the Java decoder would; the execute pass just moves bytes.
//
//   lz4_parse2_kernel     a lane per block (64 blocks per wavefront): one record per trip -- token and offset field read from the lane's
//                         LDS ring (literal bytes are jumped over, never read), the Java checks in their order, sequences cut into pieces
//                         of at most 16 literal + 16 match bytes.  Records go to 4 KiB chunks claimed from an arena with one atomic per
//                         wavefront and chunk; a block whose records do not fit is handed to the ring decoder afterwards (`only` filter).
//   seq_execute2_kernel   a wavefront per block: sx2::exec_block.
#include <type_traits>

#include "achip_seqexec.h"
#include "achip_seqexec2.h"
#include "achip_waveparse.h"

namespace achip {

// The parse pass (DESIGN 4c has the measurements).  Two things made its first version slow, neither of them
// the amount of arithmetic: (1) its lanes refilled their input windows whenever they ran dry, so that every trip some lane waited for
// a load another lane had just issued -- a wavefront waits for its memory operations in order, by count -- and the trip took one memory
// latency; (2) a dozen divergent branches per sequence.  Here the input comes through sx::LaneFeed (loads in flight for NS trips, one
// place per trip where they are requested and one where they land), and sequences that are nothing special -- no second length-extension
// byte, not within the last bytes of either buffer, offset inside the output -- take a straight-line FAST PATH: for those every Java
// check is known to pass (its conditions are exactly the complement of the Java loop's failure and last-literals branches).  Any other
// sequence is parsed by lz4_parse_general: the Java loop body restated check by check, reading the stream directly.
#if defined(__HIPCC__)
#define ACHIP_EMU_COUNT(k, cond)
#else
extern "C" long long achip_emu_counters[16];
#define ACHIP_EMU_COUNT(k, cond) achip_emu_counters[k] += (cond) ? 1 : 0
#endif
struct Lz4ParseState {
    int32_t ip, op, st, eo, litEndPrev;
    bool done, fallback;
};

// one sequence, the general way (M/lz4/Lz4RawDecompressor.java:59-195 without the copies); returns true when a record is to be emitted
__device__ __forceinline__ bool lz4_parse_general(const uint8_t* __restrict__ in, Lz4ParseState& S, int32_t inLimit, int32_t outLimit, uint32_t& rLit, uint32_t& rMl, uint32_t& rOff,
                                                  int32_t& litStart)
{
    const int32_t fastOutLimit = outLimit - 8;
#define LZ4_FAIL(detail, off)                            \
    {                                                    \
        S.st = mk_status(ACHIP_CLASS_MALFORMED, detail); \
        S.eo = (int32_t)(off);                           \
        S.done = true;                                   \
        return false;                                    \
    }
    int32_t ip = S.ip, op = S.op;
    if (ip >= inLimit) {  // the loop condition :59
        S.done = true;
        return false;
    }
    const int32_t token = (int32_t)in[ip];
    ip++;
    int32_t lit = token >> 4;  // :62-77
    if (lit == 0xF) {
        if (ip >= inLimit) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
        int32_t v;
        do {
            v = (int32_t)in[ip];
            ip++;
            lit = (int32_t)((uint32_t)lit + (uint32_t)v);
        } while (v == 255 && ip < inLimit - 15);
    }
    if (lit < 0) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
    const int64_t litEnd = (int64_t)ip + lit;
    const int64_t litOutLimit = (int64_t)op + lit;
    litStart = ip;
    rLit = (uint32_t)lit;
    rMl = 0;
    rOff = 0;
    if (litOutLimit > fastOutLimit - 4 || litEnd > inLimit - 8) {  // :82-96 last literals
        if (litOutLimit > outLimit) LZ4_FAIL(ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, ip);
        if (litEnd != inLimit) LZ4_FAIL(ACHIP_D_LZ4_INPUT_NOT_CONSUMED, ip);
        S.ip = ip + lit;
        S.op = op + lit;
        S.done = true;
        return true;
    }
    ip += lit;
    op += lit;
    const int32_t offset = (int32_t)in[ip] | ((int32_t)in[ip + 1] << 8);  // :113-119
    ip += 2;
    if (offset == 0 || offset > op) LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);
    int32_t ml = token & 0xF;  // :122-138
    if (ml == 0xF) {
        int32_t v;
        do {
            if (ip > inLimit - 5) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
            v = (int32_t)in[ip];
            ip++;
            ml = (int32_t)((uint32_t)ml + (uint32_t)v);
        } while (v == 255);
    }
    ml = (int32_t)((uint32_t)ml + 4u);
    if (ml < 0) LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
    const int64_t matchOutLimit = (int64_t)op + ml;
    if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);  // :168-171
#undef LZ4_FAIL
    rMl = (uint32_t)ml;
    rOff = (uint32_t)offset;
    S.ip = ip;
    S.op = op + ml;
    return true;
}

__global__ __launch_bounds__(64) void lz4_parse2_kernel(BatchArgs a, sx::ArenaHeader* hdr, sx::BlockMeta* meta, int32_t* only, uint64_t* arena, int32_t maxChunks, const int32_t* stats)
{
    if (stats != nullptr && lz4_pick(stats, a.nBlocks) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    constexpr int NS = 4;
    using Feed = sx::LaneFeed<NS>;
    __shared__ __attribute__((aligned(16))) uint8_t ldsIn[Feed::STRIDE * 64];
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < a.nBlocks;
    const uint8_t* in = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    const int32_t inLimit = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    Feed L;
    {
        // what lanes that request nothing read: one address per WAVEFRONT (one request per load instruction, hot in this CU's L1) -- the
        // start of the first non-empty stream of the wavefront; a single address for the whole grid would queue every wavefront of the
        // chip at one L2 channel
        const unsigned long long nonEmpty = __ballot(inLimit > 0);
        const uint8_t* anywhere = (const uint8_t*)hdr;
        if (nonEmpty != 0) {  // (uniform)
            // (an offset from the batch's base travels, not a pointer: the loads stay global_load, not flat_load)
            anywhere = a.srcBase + (int64_t)sx::shfl_u64((uint64_t)((in - a.srcBase) - (int64_t)((uintptr_t)in & 31)), __builtin_ctzll(nonEmpty));
        }
        L.init(ldsIn, lane, in, inLimit, anywhere);
    }
    Lz4ParseState S;
    S.ip = 0;
    S.op = 0;
    S.st = 0;
    S.eo = 0;
    S.litEndPrev = 0;
    S.done = !have;
    S.fallback = false;
    if (have) {
        if (inLimit == 0) {  // :48-50
            S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_LZ4_INPUT_EMPTY);
            S.done = true;
        }
        else if (outLimit == 0) {  // :52-57 (the Java method returns -1 here)
            if (!(inLimit == 1 && in[0] == 0)) {
                S.st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT);
            }
            S.done = true;
        }
    }
    const int32_t B = L.inBase;
    const int32_t fastIn = inLimit - 8;        // a literal run may end here at the latest (:82)
    const int32_t fastOut = outLimit - 8 - 4;  // a match may end here at the latest (:82, :168)

    // ---- state of the sequence under way (every flag is an int in a vector register and every common-path update a select: flags
    // kept as lane masks cost three scalar instructions each at every branch they live across -- the first version of this loop spent
    // more scalar than vector instructions) ----
    // phase 0: at its token; 1: token read (tMl0, tLit, tStart), at the offset field q; 2: parsed, its pieces are being emitted
    int32_t phase = 0;
    uint32_t tMl0 = 0;
    int32_t tLit = 0, tStart = 0, q = 0;
    int32_t sLit = 0, sLitPos = 0, sMl = 0, sOff = 0, sK = 0;  // literal bytes left, where they are, match bytes left, the offset, match pieces emitted
    int32_t sLast = 0;                                         // the block ends behind this sequence
    int32_t finished = S.done ? 1 : 0;                         // nothing more to emit
    int32_t fallback = 0;
    // ---- the record output: exactly one record per lane and trip (an empty one when there is nothing to say), eight trips to a
    // 64-byte piece, kept in registers ----
    uint64_t rec[8];
    int32_t groupAny = 0;  // this group of eight holds a record of this lane
    int32_t firstChunk = -1, chunk = -1, fill = sx::CHUNK_RECS, count = 0;

    auto trip = [&](auto tTag) {
        constexpr int T = decltype(tTag)::value;
        constexpr int SLOT = T % NS;
        L.template land<SLOT>();
        const bool active = finished == 0;
        // ---- phase 0: the token (computed by every lane from whatever its ring holds there; committed where it applies) ----
        const int32_t v0 = S.ip + B;
        const uint32_t w = L.rd32(v0);
        const uint32_t token = w & 0xFF;
        const uint32_t lit0 = token >> 4;
        const uint32_t e1 = (w >> 8) & 0xFF;
        const bool litExt = lit0 == 0xF;
        const int32_t nLit = (int32_t)(litExt ? 15u + e1 : lit0);
        const int32_t nStart = S.ip + (litExt ? 2 : 1);
        const int32_t nQ = nStart + nLit;
        const bool do0 = active && phase == 0 && L.resident(v0, 4);
        const bool gen0 = (litExt && e1 == 255) || nQ > fastIn;
        const bool ok0 = do0 && !gen0;
        tMl0 = ok0 ? (token & 0xF) : tMl0;
        tLit = ok0 ? nLit : tLit;
        tStart = ok0 ? nStart : tStart;
        q = ok0 ? nQ : q;
        phase = ok0 ? 1 : phase;
        L.restart(nQ + B, ok0 && nQ + B >= L.issueV + 64);  // a literal run that reaches well beyond everything requested: continue there
        // ---- phase 1: the offset field ----
        const int32_t v1 = q + B;
        const uint32_t x = L.rd32(v1);
        const int32_t offset = (int32_t)(x & 0xFFFF);
        const uint32_t e2 = (x >> 16) & 0xFF;
        const bool mlExt = tMl0 == 0xF;
        const int32_t ml = (int32_t)(mlExt ? 15u + e2 : tMl0) + 4;
        const int32_t opLit = S.op + tLit;
        const int32_t opEnd = opLit + ml;
        const bool do1 = active && phase == 1 && L.resident(v1, 4);
        const bool gen1 = (mlExt && e2 == 255) || offset == 0 || offset > opLit || opEnd > fastOut;
        const bool ok1 = do1 && !gen1;
        sLit = ok1 ? tLit : sLit;
        sLitPos = ok1 ? tStart : sLitPos;
        sMl = ok1 ? ml : sMl;
        sOff = ok1 ? offset : sOff;
        sK = ok1 ? 0 : sK;
        sLast = ok1 ? 0 : sLast;
        S.ip = ok1 ? q + (mlExt ? 3 : 2) : S.ip;
        S.op = ok1 ? opEnd : S.op;
        phase = ok1 ? 2 : (do1 ? 0 : phase);
        if ((do0 && gen0) || (do1 && gen1)) {  // (rare) from the token again, reading the stream directly
            uint32_t rLit = 0, rMl = 0, rOff = 0;
            int32_t rStart = 0;
            const bool emit = lz4_parse_general(in, S, inLimit, outLimit, rLit, rMl, rOff, rStart);
            sLit = (int32_t)rLit;
            sLitPos = rStart;
            sMl = (int32_t)rMl;
            sOff = (int32_t)rOff;
            sK = 0;
            sLast = S.done ? 1 : 0;
            phase = emit ? 2 : 0;
            finished = emit ? 0 : 1;  // (no record: the end of the block, or an error -- S.st)
            L.restart(S.ip + B, emit && !S.done && S.ip + B >= L.issueV + 64);
        }
        // ---- phase 2: one piece -- at most 16 literal bytes and, behind a sequence's last literal bytes, at most 16 match bytes.  A
        // match's later pieces name the largest multiple of its offset that stays inside the match's periodic source region
        // [start - offset, ..): the same bytes, but never the output of the piece before (no chains of dependent pieces) ----
        const bool do2 = finished == 0 && phase == 2;
        const int32_t pl = sLit < 16 ? sLit : 16;
        const int32_t pm = sLit > 16 ? 0 : (sMl < 16 ? sMl : 16);
        const int32_t xm = 16 * sK + sOff;
        const int32_t o = sK > 0 ? sx::largest_multiple(sOff > 0 ? sOff : 1, xm < 65535 ? xm : 65535) : sOff;
        const int32_t skip = sLitPos - S.litEndPrev;
        const bool fb = do2 && skip > sx::MAX_SKIP;  // (a gap beyond the record field: megabytes of length bytes) the ring decoder takes the block
        const bool ok2 = do2 && !fb;
        rec[T] = ok2 ? sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, (uint32_t)skip) : 0ull;
        groupAny |= ok2 ? 1 : 0;
        fallback |= fb ? 1 : 0;
        S.litEndPrev = ok2 ? sLitPos + pl : S.litEndPrev;
        sLitPos += ok2 ? pl : 0;
        sLit -= ok2 ? pl : 0;
        sMl -= ok2 ? pm : 0;
        sK += ok2 && pm > 0 ? 1 : 0;
        const bool seqEnd = ok2 && sLit == 0 && sMl == 0;
        phase = seqEnd ? 0 : phase;
        finished |= (seqEnd && sLast != 0) || fb ? 1 : 0;
        L.template issue<SLOT>((phase == 1 ? q : S.ip) + B, finished == 0);
    };

    while (__ballot(finished == 0) != 0) {  // (uniform)
        groupAny = 0;
        trip(std::integral_constant<int, 0>{});
        trip(std::integral_constant<int, 1>{});
        trip(std::integral_constant<int, 2>{});
        trip(std::integral_constant<int, 3>{});
        trip(std::integral_constant<int, 4>{});
        trip(std::integral_constant<int, 5>{});
        trip(std::integral_constant<int, 6>{});
        trip(std::integral_constant<int, 7>{});
        // ---- the group leaves: a chunk for every lane that needs one (one atomic per wavefront), then one 64-byte piece per lane ----
        const bool flush = groupAny != 0 && fallback == 0;
        const bool need = flush && fill == sx::CHUNK_RECS;
        const unsigned long long nm = __ballot(need);
        if (nm != 0) {  // (uniform)
            int32_t base = 0;
            if (lane == __builtin_ctzll(nm)) {
                base = atomicAdd(&hdr->nextChunk, (int32_t)__popcll(nm));
            }
            base = sx::wave_bcast(base, __builtin_ctzll(nm));
            if (need) {
                const int32_t c = base + (int32_t)__popcll(nm & ((1ull << lane) - 1));
                if (c >= maxChunks) {  // the arena is exhausted: the ring decoder takes the block
                    fallback = 1;
                    finished = 1;
                }
                else {
                    if (chunk >= 0) {
                        arena[(int64_t)chunk * sx::CHUNK_SLOTS + sx::CHUNK_RECS] = (uint64_t)(uint32_t)c;  // link
                    }
                    else {
                        firstChunk = c;
                    }
                    chunk = c;
                    fill = 0;
                }
            }
        }
        if (flush && fallback == 0) {
            uint8_t* const dst = (uint8_t*)(arena + (int64_t)chunk * sx::CHUNK_SLOTS + fill);
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                st16(dst + 8 * k, u32x4{(uint32_t)rec[k], (uint32_t)(rec[k] >> 32), (uint32_t)rec[k + 1], (uint32_t)(rec[k + 1] >> 32)});
            }
            fill += 8;
            count += 8;
        }
    }
    if (have) {
        if (fallback != 0) {
            only[block] = 1;
            meta[block].firstChunk = 0;
            meta[block].count = 0;
            atomicAdd(&hdr->fallbackBlocks, 1);
        }
        else {
            only[block] = 0;
            meta[block].firstChunk = firstChunk < 0 ? 0 : firstChunk;
            meta[block].count = S.st == 0 ? count : 0;
            a.outLen[block] = S.st == 0 ? S.op : 0;
            a.status[block] = S.st;
            a.errOffset[block] = (int64_t)S.eo;
        }
    }
}

// ---- lz4_parse_wave_kernel: the parse pass with a WAVEFRONT per block (round 4), for batches of FEW blocks.
// The lane-per-block parser above is the efficient one -- 64 blocks per instruction stream -- when there are blocks for every lane of the chip
// (262 144 and up).  The stream readers hand over a few thousand LARGE blocks (an LZ4 frame's blocks of up to 4 MiB, a Hadoop stream's chunks of
// 256 KiB): a few wavefronts, each alone on its SIMD, each lane a serial chain of one trip (~1 500 cycles) per sequence -- a 256 KiB chunk took
// 15 ms, a 4 MiB block 250, whatever the rest of the chip did.  Here the wavefront parses ONE block, 64 token positions per trip:
//   * the stream passes through a staging area in LDS (WaveStage); lane p reads the window's bytes AS IF a sequence started at position p: token, literal length (one
//     extension byte at most), where the offset field would be, match length (likewise), and where the sequence after it would begin;
//   * the real sequences of the window are the chain 0 -> next[0] -> next[next[0]] ...: a scalar loop of lane reads (a handful of scalar
//     instructions per sequence instead of a trip);
//   * the lanes on the chain get their output positions from a scan, make the Java loop's checks (exactly the fast path's of the parser above:
//     its conditions are the complement of the Java loop's failure and last-literals branches) and store one record each.
// A sequence that is anything else -- a second extension byte, more than 16 literal or match bytes (several records), a failing check, the last
// 360 bytes of the block -- ends the chain in front of it and is parsed by lz4_parse_general, the Java loop body check by check, its records
// written by the whole wavefront (a literal run of megabytes is 64 records per step).  Same records, same statuses and error offsets as the
// parser above: the executor does not know which of the two wrote them.

__global__ __launch_bounds__(64) void lz4_parse_wave_kernel(BatchArgs a, sx::ArenaHeader* hdr, sx::BlockMeta* meta, int32_t* only, uint64_t* arena, int32_t maxChunks, const int32_t* stats)
{
    if (stats != nullptr && lz4_pick(stats, a.nBlocks) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    __shared__ __attribute__((aligned(16))) uint8_t stageLds[WaveStage<wp::LZ4_STAGE>::CAP + 16];
    const int lane = threadIdx.x;
    const int64_t block = blockIdx.x;
    const uint8_t* __restrict__ in = a.srcBase + a.srcOff[block];
    const int32_t inLimit = uni(a.srcLen[block]);
    WaveStage<wp::LZ4_STAGE> W;
    W.lds = stageLds;
    W.in = in;
    W.inLimit = inLimit;
    W.b0 = -1;
    W.pend[0] = u32x4{0, 0, 0, 0};
    W.pend[1] = u32x4{0, 0, 0, 0};
    W.lane = lane;
    const int32_t outLimit = uni(a.dstCap[block]);
    Lz4ParseState S;
    S.ip = 0;
    S.op = 0;
    S.st = 0;
    S.eo = 0;
    S.litEndPrev = 0;
    S.done = false;
    S.fallback = false;
    if (inLimit == 0) {  // :48-50
        S.st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_LZ4_INPUT_EMPTY);
        S.done = true;
    }
    else if (outLimit == 0) {  // :52-57 (the Java method returns -1 here)
        if (!(inLimit == 1 && in[0] == 0)) {
            S.st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT);
        }
        S.done = true;
    }
    WaveRecordSink K;
    K.hdr = hdr;
    K.arena = arena;
    K.maxChunks = maxChunks;
    K.firstChunk = -1;
    K.chunk = -1;
    K.fill = sx::CHUNK_RECS;
    K.count = 0;
    K.fallback = false;
    K.fresh = -1;
    const int32_t fastOut = outLimit - 8 - 4;  // a match may end here at the latest (:82, :168)
    bool finished = S.done;                    // (uniform)
    bool serial = false;                       // (uniform) the sequences are long: one at a time (below)
    while (!finished && !K.fallback) {         // (uniform)
        bool general = true;
        const bool windowable = (int64_t)S.ip + wp::LZ4_STAGE + 24 <= (int64_t)inLimit;  // (uniform) nothing a window looks at can reach the block's last bytes (8: the Java loop's margin; 16: every byte it looks at lies in a 16-byte piece that is inside the block whole -- WaveStage::fetch)
        if (windowable && serial) {
            // LONG SEQUENCES, one at a time.  A window of 64 stream positions finds as many sequences as start in it: on data whose sequences are longer than the window
            // (runs of literals, long matches: the headline's kind) that is ONE per trip of ~2 800 clocks.  Here every lane reads the sequence at the window's first
            // position -- two LDS reads, no chain, no scans --, lane k makes its piece k.  The mode is entered when a window held at most two sequences in 48 bytes and
            // left when a sequence is shorter than 24 stream bytes.  Same conditions as a member of the chain; anything else goes the general way.
            const int32_t base = S.ip;
            const uint8_t* const stage = W.window(base);
            uint32_t x;
            __builtin_memcpy(&x, stage, 4);
            const uint32_t token = x & 0xFF, e1 = (x >> 8) & 0xFF;
            const bool litExt = (token >> 4) == 0xF;
            const int32_t lit = uni((int32_t)(litExt ? 15u + e1 : (token >> 4)));
            const int32_t litStart = litExt ? 2 : 1;
            const int32_t q = litStart + lit;
            uint32_t y;
            __builtin_memcpy(&y, stage + q, 4);
            const int32_t offset = uni((int32_t)(y & 0xFFFF));
            const uint32_t e2 = (y >> 16) & 0xFF;
            const bool mlExt = (token & 0xF) == 0xF;
            const int32_t ml = uni((int32_t)(mlExt ? 15u + e2 : (token & 0xF)) + 4);
            const int32_t next = q + (mlExt ? 3 : 2);
            const int32_t opLit = S.op + lit, opEnd = opLit + ml;
            const int32_t skip = base + litStart - S.litEndPrev;
            const bool ok = !((litExt && e1 == 255) || (mlExt && e2 == 255)) && offset != 0 && offset <= opLit && opEnd <= fastOut && skip <= sx::MAX_SKIP;
            serial = false;
            if (uni(ok ? 1 : 0) != 0) {  // (uniform)
                const int32_t litFull = lit > 16 ? (lit + 15) / 16 - 1 : 0;
                const int32_t matchRest = ml > 16 ? (ml - 16 + 15) / 16 : 0;
                const int32_t pieces = litFull + 1 + matchRest;  // (<= 35)
                const int32_t k = lane;
                int32_t pl, pm, o = offset;
                if (k < litFull) {
                    pl = 16;
                    pm = 0;
                }
                else if (k == litFull) {
                    pl = lit - 16 * litFull;
                    pm = ml < 16 ? ml : 16;
                }
                else {
                    const int32_t m = k - litFull;
                    pl = 0;
                    pm = ml - 16 * m < 16 ? ml - 16 * m : 16;
                    const int32_t xm = 16 * m + offset;
                    o = sx::largest_multiple(offset, xm < 65535 ? xm : 65535);
                }
                K.put(sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, k == 0 ? (uint32_t)skip : 0u), k < pieces, lane, pieces, lane);
                if (!K.fallback) {
                    S.op = opEnd;
                    S.litEndPrev = base + q;
                    S.ip = base + next;
                    general = false;
                    serial = next >= 24;
                }
            }
        }
        else if (windowable) {
            const int32_t base = S.ip;
            const uint8_t* const stage = W.window(base);
            // what a sequence at position `lane` of the window would be
            uint32_t x;
            __builtin_memcpy(&x, stage + lane, 4);
            const uint32_t token = x & 0xFF, e1 = (x >> 8) & 0xFF;
            const bool litExt = (token >> 4) == 0xF;
            const int32_t lit = (int32_t)(litExt ? 15u + e1 : (token >> 4));
            const int32_t litStart = lane + (litExt ? 2 : 1);
            const int32_t q = litStart + lit;  // the offset field (<= 334)
            uint32_t y;
            __builtin_memcpy(&y, stage + q, 4);
            const int32_t offset = (int32_t)(y & 0xFFFF);
            const uint32_t e2 = (y >> 16) & 0xFF;
            const bool mlExt = (token & 0xF) == 0xF;
            const int32_t ml = (int32_t)(mlExt ? 15u + e2 : (token & 0xF)) + 4;
            const int32_t next = q + (mlExt ? 3 : 2);
            // not for the chain: a second extension byte
            const bool stop = (litExt && e1 == 255) || (mlExt && e2 == 255);
            const unsigned long long stopMask = __ballot(stop);
            unsigned long long members = 0;
            int32_t cur = 0;
            wave_chain(next, stop, stopMask, lane, members, cur);
            // the members' places in the output, and the checks that need them (:113-119, :168-171 as the fast path above has them)
            const bool member = ((members >> lane) & 1ull) != 0;
            // (a sequence of more than 16 literal or match bytes is several records -- round 5: until then such a sequence ended the chain and went through
            // lz4_parse_general, a sixth of a text block's sequences, each a few dependent reads of global memory -- : pieces as in the general path below)
            const int32_t litFull = lit > 16 ? (lit + 15) / 16 - 1 : 0;
            const int32_t matchRest = ml > 16 ? (ml - 16 + 15) / 16 : 0;
            int32_t pieces = litFull + 1 + matchRest;
            // one scan for both: output bytes in the low half (at most 64 x 542), pieces in the high half (at most 64 x 35)
            const int32_t scanned = sx::wave_scan_incl(member ? ((lit + ml) | (pieces << 16)) : 0, lane);
            const int32_t endRel = scanned & 0xFFFF, pieceEnd = scanned >> 16;
            const int32_t opEnd = S.op + endRel, opLit = opEnd - ml;
            // a failing check ends the chain in front of the sequence; and a window's records reach into one new chunk at most, so the chain also ends where
            // they would exceed a chunk (a member has at most 35 pieces: the first always fits)
            const bool wrong = member && (offset == 0 || offset > opLit || opEnd > fastOut || pieceEnd > sx::CHUNK_RECS);
            const unsigned long long wrongMask = __ballot(wrong);
            if (wrongMask != 0) {  // (uniform)
                const int first = __builtin_ctzll(wrongMask);
                members &= (1ull << first) - 1ull;
                cur = first;
            }
            const bool mine = ((members >> lane) & 1ull) != 0;
            pieces = mine ? pieces : 0;
            if (members != 0) {  // (uniform)
                const unsigned long long below = members & ((1ull << lane) - 1ull);
                const int prevLane = below != 0 ? 63 - __builtin_clzll(below) : 0;
                const int32_t prevQ = __shfl(q, prevLane);
                const int32_t skip = base + litStart - (below != 0 ? base + prevQ : S.litEndPrev);
                const int last = 63 - __builtin_clzll(members);
                const int32_t n = sx::wave_bcast(pieceEnd, last);
                if (__ballot(mine && skip > sx::MAX_SKIP) != 0) {  // (a gap beyond the record field) the ring decoder takes the block
                    K.fallback = true;
                }
                else if (K.begin(n, lane)) {
                    for (int32_t k = 0; __ballot(k < pieces) != 0; k++) {  // (uniform) piece k of every sequence that has one: one or two rounds on text
                        int32_t pl, pm, o = offset;
                        if (k < litFull) {
                            pl = 16;
                            pm = 0;
                        }
                        else if (k == litFull) {
                            pl = lit - 16 * litFull;
                            pm = ml < 16 ? ml : 16;
                        }
                        else {
                            const int32_t m = k - litFull;  // match pieces before this one
                            pl = 0;
                            pm = ml - 16 * m < 16 ? ml - 16 * m : 16;
                            const int32_t xm = 16 * m + offset;
                            o = sx::largest_multiple(offset > 0 ? offset : 1, xm < 65535 ? xm : 65535);
                        }
                        K.store(sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, k == 0 ? (uint32_t)skip : 0u), k < pieces, pieceEnd - pieces + k);
                    }
                    K.end(n);
                    S.op += sx::wave_bcast(endRel, last);
                    S.litEndPrev = base + sx::wave_bcast(q, last);
                    S.ip = base + cur;
                    general = false;
                    serial = cur >= 48 && __popcll(members) <= 2;
                }
            }
        }
        if (general && !K.fallback) {  // (uniform) one sequence the Java way, its records by the whole wavefront
            uint32_t rLit = 0, rMl = 0, rOff = 0;
            int32_t rStart = 0;
            const bool emit = lz4_parse_general(in, S, inLimit, outLimit, rLit, rMl, rOff, rStart);
            if (emit) {
                const int32_t sLit = (int32_t)rLit, sMl = (int32_t)rMl, sOff = (int32_t)rOff;
                const int32_t litFull = sLit > 16 ? (sLit + 15) / 16 - 1 : 0;
                const int32_t matchRest = sMl > 16 ? (sMl - 16 + 15) / 16 : 0;
                const int32_t pieces = litFull + 1 + matchRest;
                const int32_t skip0 = rStart - S.litEndPrev;
                if (skip0 > sx::MAX_SKIP) {
                    K.fallback = true;
                }
                for (int32_t k0 = 0; k0 < pieces && !K.fallback; k0 += 64) {  // (uniform)
                    const int32_t k = k0 + lane;
                    int32_t pl, pm, o = sOff;
                    if (k < litFull) {
                        pl = 16;
                        pm = 0;
                    }
                    else if (k == litFull) {
                        pl = sLit - 16 * litFull;
                        pm = sMl < 16 ? sMl : 16;
                    }
                    else {
                        const int32_t m = k - litFull;  // match pieces before this one
                        pl = 0;
                        pm = sMl - 16 * m < 16 ? sMl - 16 * m : 16;
                        const int32_t xm = 16 * m + sOff;
                        o = sx::largest_multiple(sOff > 0 ? sOff : 1, xm < 65535 ? xm : 65535);
                    }
                    const int32_t left = pieces - k0;
                    K.put(sx::rec_pack((uint32_t)pl, (uint32_t)pm, pm > 0 ? (uint32_t)o : 0u, k == 0 ? (uint32_t)skip0 : 0u), k < pieces, lane, left < 64 ? left : 64, lane);
                }
                S.litEndPrev = rStart + sLit;
            }
            finished = !emit || S.done;
        }
    }
    if (lane == 0) {
        if (K.fallback) {
            only[block] = 1;
            meta[block].firstChunk = 0;
            meta[block].count = 0;
            atomicAdd(&hdr->fallbackBlocks, 1);
        }
        else {
            only[block] = 0;
            meta[block].firstChunk = K.firstChunk < 0 ? 0 : K.firstChunk;
            meta[block].count = S.st == 0 ? K.count : 0;
            a.outLen[block] = S.st == 0 ? S.op : 0;
            a.status[block] = S.st;
            a.errOffset[block] = (int64_t)S.eo;
        }
    }
}

// which of the two parsers: 0 = by the batch (a wavefront per block up to 16 384 blocks with a count known to the host, a lane per block above), 1 = a lane per block,
// 2 = a wavefront per block (context option lz4.decompress.parse).  The lane parser's time does not grow with the count until the chip is full (64 blocks per instruction
// stream), the wavefront parser's does from 8 192 blocks on (the wavefronts a launch has resident).  Corpus blocks of 64 KiB, GiB/s lane / wavefront (profiles/r05_parsesweep.txt):
// 4 096 blocks 48 / 132; 8 192: 90 / 153; 16 384: 163 / 186; 32 768: 255 / 205; 131 072: 443 / 220.  (Round 4 had the line at 4 096: the wavefront parser was half as fast then.)
constexpr int32_t LZ4_WAVE_PARSE_MAX_BLOCKS = 16384;
int g_lz4_parse_mode = 0;

// the execute pass (achip_seqexec2.h): pieces of at most 16 + 16 bytes, every global load one batch ahead
template <int WIN = sx2::WIN_DEFAULT, int WAVES = 0>
__global__ __launch_bounds__(64, WAVES) void seq_execute2_kernel(BatchArgs a, const sx::BlockMeta* meta, const uint64_t* arena, const int32_t* stats, int32_t shortLimit)
{
    if (stats != nullptr && lz4_pick(stats, a.nBlocks, shortLimit) != LZ4_PICK_TWOPASS) {  // auto mode: the ring decoder takes this batch
        return;
    }
    __shared__ __attribute__((aligned(16))) uint8_t win[WIN + 16];
    const int64_t block = blockIdx.x;
    const sx::BlockMeta m = meta[block];
    if (m.count <= 0) {
        return;
    }
    sx2::exec_block<WIN>(win, a.srcBase + a.srcOff[block], a.srcLen[block], a.dstBase + a.dstOff[block], arena, m.firstChunk, m.count, (int)threadIdx.x);
}

// scratch: [header 256 B][meta n x 8][only n x 4][arena, 4 KiB aligned]
// (perBlock: arena bytes per block -- 8 bytes per record.  Pieces of <= 16 + 16 bytes: text-like 64 KiB blocks make 6 000 .. 8 500 LZ4 records and
// 8 500 .. 11 500 Snappy records, a few percent of them empty ones from trips a lane sat out; blocks that do not fit go to the ring decoder.)
int64_t twopass_scratch_bytes(int32_t nBlocks, int64_t perBlock)
{
    const int64_t fixed = 4096 + (((int64_t)nBlocks * 12 + 4095) & ~4095LL);
    int64_t arena = (int64_t)nBlocks * perBlock + (64LL << 20) + 4096;  // (+ the chunk the executor's record loads may run into)
    return fixed + arena;
}
int64_t lz4_twopass_scratch_bytes(int32_t nBlocks) { return twopass_scratch_bytes(nBlocks, 98304); }

// the execute pass (shared with snappy_decompress_v5.hip).  (execVariant: 2, the only one -- round 2's timing aids, executor and parser variants
// that left work out, were development tools and went in round 4 together with their build switch)
hipError_t launch_seq_execute2(const BatchArgs& a, hipStream_t stream, const sx::BlockMeta* meta, const uint64_t* arena, int execVariant, const int32_t* stats, int32_t shortLimit)
{
    (void)execVariant;
    hipLaunchKernelGGL(seq_execute2_kernel<>, dim3((unsigned)a.nBlocks), dim3(64), 0, stream, a, meta, arena, stats, shortLimit);
    return hipGetLastError();
}

hipError_t launch_lz4_decompress_rings(const BatchArgs& a, hipStream_t stream, int groupSize, int ringClass, const int32_t* mixedGroups);

// The two passes of a batch on the caller's stream.  (Round 2 left an experiment here -- the batch cut into 2 .. 8 parts alternating between two
// helper streams, so that one part's parse could share the chip with another part's execute; measured in round 3 on the corpus batch
// (profiles/r03_notes.md): 2 parts 520 GiB/s against 517, 4 parts 456, 8 parts 363: the two kernels compete for the same issue slots.
// Removed, with the 8 KiB-window executor (489) and the register-capped one.)
hipError_t launch_lz4_decompress_twopass(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int groupSize, int ringClass, int execVariant, const int32_t* stats)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* s = (uint8_t*)scratch;
    sx::ArenaHeader* hdr = (sx::ArenaHeader*)s;
    sx::BlockMeta* meta = (sx::BlockMeta*)(s + 4096);
    int32_t* only = (int32_t*)(s + 4096 + (int64_t)a.nBlocks * 8);
    const int64_t fixed = 4096 + (((int64_t)a.nBlocks * 12 + 4095) & ~4095LL);
    uint64_t* arena = (uint64_t*)(s + fixed);
    const int64_t chunks = (scratchBytes - fixed) / (sx::CHUNK_SLOTS * 8) - 1;  // (one to spare: the executor's unconditional record loads)
    const int32_t maxChunks = (int32_t)(chunks > 0x7FFFFFFF ? 0x7FFFFFFF : chunks);
    hipError_t e = hipMemsetAsync(hdr, 0, sizeof(sx::ArenaHeader), stream);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((a.nBlocks + 63) / 64)), wg(64);
    {
    const bool wavePerBlock = a.nBlocksDev == nullptr && (g_lz4_parse_mode == 2 || (g_lz4_parse_mode == 0 && a.nBlocks <= LZ4_WAVE_PARSE_MAX_BLOCKS));
    if (wavePerBlock) {
        hipLaunchKernelGGL(lz4_parse_wave_kernel, dim3((unsigned)a.nBlocks), wg, 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
    }
    else {
        hipLaunchKernelGGL(lz4_parse2_kernel, grid, wg, 0, stream, a, hdr, meta, only, arena, maxChunks, stats);
    }
    e = launch_seq_execute2(a, stream, meta, arena, execVariant, stats, 12);
    if (e != hipSuccess) return e;
    }
    BatchArgs f = a;
    f.only = only;
    f.onlyStats = stats;
    f.onlyShortLimit = 12;
    e = launch_lz4_decompress_rings(f, stream, groupSize, ringClass, nullptr);
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace achip
