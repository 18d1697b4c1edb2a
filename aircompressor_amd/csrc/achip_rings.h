// achip_rings.h -- LDS ring buffers for the streaming LZ77 decoders (v2 kernels).
//
// Every block (one GS-lane group of a wavefront) owns two rings in LDS:
//   * an INPUT ring: the compressed stream is pulled from HBM exactly once, in 16-byte aligned
//     granules, GS granules (GS*16 contiguous bytes) per refill -- coalesced, no partial lines;
//   * an OUTPUT ring: the sliding history window.  Literal and match bytes are produced into it
//     one byte per lane per step; completed GS*16-byte aligned chunks are flushed to HBM with one
//     16-byte store per lane -- every output line is written once, whole.
// Back-references that reach at most LDS_REACH bytes back are served from the output ring (LDS
// latency, no memory traffic); farther ones re-read the block's own flushed output through L2.
// All positions are "virtual": absolute position + (buffer address & 15), so that virtual 0 is a
// 16-byte aligned address; ring index = virtual position & (RING-1).
#pragma once
#include "achip_device.h"

namespace achip {

// Ring pairs of consecutive blocks are BatchArgs::ringPad bytes apart beyond their size (a multiple of 16): with a
// 384 B = 96-dword slot the 16 blocks of a wavefront start on only two distinct LDS banks; 400 B = 100 dwords
// spreads them over all banks (context option "decompress.ring_pad").

// PHASED (round 3): ONE place per sequence where this group's vector memory operations are issued -- memory_phase(): the input chunk
// requested a sequence ago enters the ring, completed output chunks leave, the next input chunk is requested -- instead of refills and
// flushes wherever a copy happens to need them.  Why: a wavefront waits for its vector memory operations by COUNT, and with loads and
// stores under data-dependent branches the compiler can only wait for all of them (`s_waitcnt vmcnt(0)`, 49 of the 52 vmcnt waits of the
// round-2 kernel): every refill waited for the stores of the flush just before it to be acknowledged (SQ_WAIT_ANY 57 % of the
// wave-cycles, profiles/r03_notes.md).  With all of a sequence's memory operations in one place, whatever a wait covers is a whole
// sequence old.  The copies then only flush inside loops (long runs), the input ring is kept two chunks ahead (IN_RING >= 4 chunks).
template <int GS, int IN_RING, int OUT_RING, int GPL = 1, int PHASED = 0>  // PHASED != 0: the decode loops top the input ring up once per trip (memory_phase)
struct Rings {
    static constexpr int CHUNK = GS * 16 * GPL;                // bytes per refill / flush / copy step (GPL 16-byte granules per lane)
    static constexpr int LDS_REACH = OUT_RING - CHUNK - 16;    // farthest back-reference served from the ring
    static_assert((IN_RING & (IN_RING - 1)) == 0 && (OUT_RING & (OUT_RING - 1)) == 0, "rings are powers of two");
    static_assert(IN_RING >= 2 * CHUNK && OUT_RING >= 4 * CHUNK, "ring too small for the chunk size");
    static_assert(!PHASED || (IN_RING >= 4 * CHUNK && GS == 4 && GPL == 1), "the phased form keeps two chunks of input ahead");
    static constexpr bool PH_REFILL = PHASED != 0, PH_FLUSH = false;  // (PH_FLUSH: the flushes deferred to the phase as well -- measured 3 .. 11 % slower, profiles/r03_notes.md; the code paths stay for the record of what was measured)

    uint8_t* inRing;
    uint8_t* outRing;
    const uint8_t* inAligned;  // in - inBase
    uint8_t* outAligned;       // out - outBase
    int32_t inBase, outBase;
    int32_t inEndV;            // virtual end of the input
    int32_t inLoadedV;         // input ring holds virtual [inLoadedV - IN_RING, inLoadedV)
    int32_t flushedV;          // output flushed to HBM up to this virtual position (multiple of 16, or the final end)
    u32x4 pending[GPL];        // this lane's granules of the NEXT input chunk, requested one refill ahead (hides HBM latency)
    int g;
    uint8_t* stage;            // CHUNK bytes of LDS behind the rings (or null): landing zone of back-references read from HBM

    // GS == 4: ONE generic <= CHUNK-byte move per loop trip, whatever its length and wherever it comes from.  The 16
    // blocks of a wavefront diverge on every special case (short / long, near / far, overlapping), and a wavefront pays
    // for every path any of its blocks takes: on text that was ~785 wave instructions per 16 sequences.
    static constexpr bool UNIFIED = GS == 4 && GPL == 1;

    // The lanes of a group hand bytes to each other through the rings.  On the device they run in lockstep and a wavefront's memory
    // operations are performed in order: order() only has to keep the compiler from moving loads over stores, enter() is nothing.
    // tools/hostemu runs lanes as fibers: with HOSTEMU_RINGS_LOCKSTEP both become a rendezvous of the GROUP, which restores what the
    // code relies on -- nobody reads before everybody has written (order), nobody writes before everybody has read (enter).
#if !defined(__HIPCC__) && defined(HOSTEMU_RINGS_LOCKSTEP)
    __device__ __forceinline__ void order() const { if (GS > 1) hostemu::group_sync(GS, __FILE__, __LINE__); }
    __device__ __forceinline__ void enter() const { if (GS > 1) hostemu::group_sync(GS, __FILE__, __LINE__); }
#elif !defined(__HIPCC__) && defined(HOSTEMU_RINGS_ORDER_ONLY)
    // (tools/hostemu/emu_lockstep.cpp: under access-granular lockstep the group only has to meet where the device has its compiler barrier)
    __device__ __forceinline__ void order() const { if (GS > 1) hostemu::group_sync(GS, __FILE__, __LINE__); }
    __device__ __forceinline__ void enter() const {}
#else
    __device__ __forceinline__ void order() const { wave_mem_order(); }
    __device__ __forceinline__ void enter() const {}
#endif

    __device__ __forceinline__ void init(uint8_t* ldsIn, uint8_t* ldsOut, const uint8_t* in, int32_t inLimit, uint8_t* out, int lane, uint8_t* ldsStage = nullptr)
    {
        inRing = ldsIn;
        outRing = ldsOut;
        stage = ldsStage;
        inBase = (int32_t)((uintptr_t)in & 15);
        outBase = (int32_t)((uintptr_t)out & 15);
        inAligned = in - inBase;
        outAligned = out - outBase;
        inEndV = inLimit + inBase;
        inLoadedV = 0;
        flushedV = 0;
        g = lane;
        if (PHASED) {
            request_pending(16 * g);
        }
        else {
#pragma unroll
            for (int q = 0; q < GPL; q++) {
                pending[q] = fetch_granule(16 * (g + GS * q));
            }
        }
    }

    // (Round 3 tried to hide the prefetch from the compiler -- the load issued through inline assembly, waited for by hand -- so that none
    // of its conservative waits would cover it.  tools/check_hidden_loads.py, a data-flow check of the generated code, showed why not: the
    // compiler lands the asm's result in temporaries and copies it home, reading registers with the load in flight.  What works instead is
    // to keep every wait the compiler places out of the common path: see ensure_input.)
    __device__ __forceinline__ void request_pending(int32_t v) { pending[0] = fetch_granule(v); }
    __device__ __forceinline__ u32x4 take_pending(int32_t) { return pending[0]; }

    // switch the input ring to a new source stream (Zstd: literals of the next block, a raw block, ...)
    __device__ __forceinline__ void reset_input(const uint8_t* in, int32_t inLimit)
    {
        enter();
        inBase = (int32_t)((uintptr_t)in & 15);
        inAligned = in - inBase;
        inEndV = inLimit + inBase;
        inLoadedV = 0;
        order();
#pragma unroll
        for (int q = 0; q < GPL; q++) {
            pending[q] = fetch_granule(16 * (g + GS * q));
        }
    }

    // this lane's 16-byte granule at virtual position v; bytes outside the input read as 0
    __device__ __forceinline__ u32x4 fetch_granule(int32_t v) const
    {
        u32x4 d = {0, 0, 0, 0};
        if (v >= inBase && v + 16 <= inEndV) {
            d = ld16_global(inAligned + v);  // aligned, fully inside the input: one coalesced CHUNK per group (a GLOBAL load: a flat one also counts as an LDS operation, and every LDS wait then waits for memory)
        }
        else if (v + 16 > inBase && v < inEndV) {  // first / last granule of the stream: byte-guarded, kept small (cold)
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
            for (int i = 0; i < 16; i++) {
                const int32_t p = v + i;
                if (p >= inBase && p < inEndV) {
                    w[i >> 2] |= (uint32_t)inAligned[p] << (8 * (i & 3));
                }
            }
            d = u32x4{w[0], w[1], w[2], w[3]};
        }
        return d;
    }

    // ---- input side ----
    __device__ __forceinline__ void refill()
    {
        if (PHASED) {
            const int32_t v = inLoadedV + 16 * g;
            *(u32x4*)(inRing + (v & (IN_RING - 1))) = take_pending(v);  // requested during the previous refill
            request_pending(v + CHUNK);
            inLoadedV += CHUNK;
            return;
        }
#pragma unroll
        for (int q = 0; q < GPL; q++) {
            const int32_t v = inLoadedV + 16 * (g + GS * q);
            *(u32x4*)(inRing + (v & (IN_RING - 1))) = pending[q];  // requested during the previous refill
            pending[q] = fetch_granule(v + CHUNK);
        }
        inLoadedV += CHUNK;
    }
    // make input bytes [pos, pos+need) readable from the ring (need <= CHUNK); bytes past the input end read as 0
    __device__ __forceinline__ void ensure_input(int32_t pos, int32_t need)
    {
        enter();
        const int32_t want = pos + inBase + need;
        // (an `if` around the loop, not a bare `while`: the compiler flushes the memory counters in the PREHEADER of a loop that uses a
        // register with a load in flight -- `s_waitcnt vmcnt(0)` on every pass through here, refill or not, i.e. a wait for the chunk
        // requested a moment ago at each of a sequence's four calls -- whereas a wait inside a branch is only paid when it is taken)
        if (want > inLoadedV && inLoadedV < inEndV) {
            do {
                refill();
            } while (want > inLoadedV && inLoadedV < inEndV);
        }
        order();
    }
    __device__ __forceinline__ uint32_t in_u8(int32_t pos) const { return inRing[(pos + inBase) & (IN_RING - 1)]; }

    // PHASED: the one place per sequence (the decode loops call it at the top of a trip; ip / op = bytes consumed / produced so far)
    __device__ __forceinline__ void memory_phase(int32_t ip, int32_t op)
    {
        if (PHASED) {
            enter();
            // the chunk requested one phase ago enters the ring as soon as what it overwrites is consumed (the only wait: everything in
            // flight is a sequence old), the next one is requested, completed output chunks leave
            if (PH_REFILL && inLoadedV - (ip + inBase) <= IN_RING - CHUNK && inLoadedV < inEndV) {
                refill();
            }
            order();
            if (PH_FLUSH) {
                flush_complete(op);
            }
        }
    }

    // 4 bytes at an arbitrary virtual position of a ring: two aligned LDS dword reads + v_alignbyte
    template <int RING>
    static __device__ __forceinline__ uint32_t ring_ld4(const uint8_t* ring, int32_t pv)
    {
        const int32_t a = pv & ~3;
        const uint32_t lo = *(const uint32_t*)(ring + (a & (RING - 1)));
        const uint32_t hi = *(const uint32_t*)(ring + ((a + 4) & (RING - 1)));
        return alignbyte_u32(hi, lo, (uint32_t)(pv & 3));
    }


    // c bytes (1 <= c <= CHUNK) from virtual position sV of ring `src` to virtual position dV of the output ring.  The
    // <= 3 bytes up to the next destination dword go as bytes; from there every lane moves its CHUNK / GS bytes as
    // aligned destination dwords WHATEVER c is -- what lands beyond dV + c is scratch space of the ring (LDS_REACH leaves
    // CHUNK + 16 bytes for it) that the next move overwrites, so the move has no length-dependent predicates and no tail.
    // The source must not overlap the c destination bytes.
    template <int SRC_RING>
    __device__ __forceinline__ void copy_dwords(const uint8_t* src, int32_t sV, int32_t dV, int32_t c)
    {
        copy_dwords_rt(src, SRC_RING - 1, sV, dV, c);
    }

    // Short copies (n <= 4*GS, source entirely before the destination): lane g moves bytes [4g, 4g+4) -- one unaligned
    // 4-byte read, up to four byte stores, no alignment cases.  Text-like data is almost all such copies.
    __device__ __forceinline__ void put4(int32_t dV, uint32_t w, int32_t cnt)
    {
        if (cnt > 0) outRing[dV & (OUT_RING - 1)] = (uint8_t)w;
        if (cnt > 1) outRing[(dV + 1) & (OUT_RING - 1)] = (uint8_t)(w >> 8);
        if (cnt > 2) outRing[(dV + 2) & (OUT_RING - 1)] = (uint8_t)(w >> 16);
        if (cnt > 3) outRing[(dV + 3) & (OUT_RING - 1)] = (uint8_t)(w >> 24);
    }
    template <int SRC_RING>
    __device__ __forceinline__ void copy_small(const uint8_t* src, int32_t sV, int32_t dV, int32_t n)
    {
        const uint32_t w = ring_ld4<SRC_RING>(src, sV + 4 * g);
        enter();
        put4(dV + 4 * g, w, n - 4 * g);
    }

    // copy_dwords with the source ring's size given at run time (mask = size - 1): lets one instruction stream serve
    // literal runs (input ring), near matches (output ring) and staged far matches alike -- see the *_steps kernels
    static __device__ __forceinline__ uint32_t ring_ld4_rt(const uint8_t* ring, int32_t mask, int32_t pv)
    {
        const int32_t a = pv & ~3;
        const uint32_t lo = *(const uint32_t*)(ring + (a & mask));
        const uint32_t hi = *(const uint32_t*)(ring + ((a + 4) & mask));
        return alignbyte_u32(hi, lo, (uint32_t)(pv & 3));
    }
    __device__ __forceinline__ void copy_dwords_rt(const uint8_t* src, int32_t mask, int32_t sV, int32_t dV, int32_t c)
    {
        int32_t head = (4 - (dV & 3)) & 3;
        head = head < c ? head : c;
        constexpr int ITER = CHUNK / (4 * GS);
        uint32_t w[ITER];
#pragma unroll
        for (int q = 0; q < ITER; q++) {
            w[q] = ring_ld4_rt(src, mask, sV + head + 4 * (g + GS * q));
        }
        uint32_t hb[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; k++) {  // head byte k is lane (k mod GS)'s
            if ((k & (GS - 1)) == g && k < head) {
                hb[k] = src[(sV + k) & mask];
            }
        }
        enter();  // (every lane has read before any lane writes: on the device the loads above are issued before the stores below)
#pragma unroll
        for (int q = 0; q < ITER; q++) {
            *(uint32_t*)(outRing + ((dV + head + 4 * (g + GS * q)) & (OUT_RING - 1))) = w[q];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if ((k & (GS - 1)) == g && k < head) {
                outRing[(dV + k) & (OUT_RING - 1)] = (uint8_t)hb[k];
            }
        }
    }

    // ---- output side ----
    __device__ __forceinline__ void out_put(int32_t pos, uint32_t byte) { outRing[(pos + outBase) & (OUT_RING - 1)] = (uint8_t)byte; }
    __device__ __forceinline__ uint32_t out_get(int32_t pos) const { return outRing[(pos + outBase) & (OUT_RING - 1)]; }

    // flush every complete CHUNK below position `op` (absolute)
    __device__ __forceinline__ void flush_complete(int32_t op)
    {
        enter();
        const int32_t opV = op + outBase;
        order();
        while (flushedV + CHUNK <= opV) {
#pragma unroll
            for (int q = 0; q < GPL; q++) {
                const int32_t v = flushedV + 16 * (g + GS * q);
                if (v >= outBase) {
                    *(u32x4*)(outAligned + v) = *(const u32x4*)(outRing + (v & (OUT_RING - 1)));
                }
                else if (v + 16 > outBase) {  // the granule straddling the start of the output buffer
                    for (int32_t p = outBase; p < v + 16; p++) {
                        outAligned[p] = outRing[p & (OUT_RING - 1)];
                    }
                }
            }
            flushedV += CHUNK;
        }
        order();
    }
    // flush everything up to `op` (end of block)
    __device__ __forceinline__ void flush_all(int32_t op)
    {
        flush_complete(op);
        const int32_t opV = op + outBase;
#pragma unroll
        for (int q = 0; q < GPL; q++) {
            const int32_t v = flushedV + 16 * (g + GS * q);
            if (v < opV) {
                if (v >= outBase && v + 16 <= opV) {
                    *(u32x4*)(outAligned + v) = *(const u32x4*)(outRing + (v & (OUT_RING - 1)));
                }
                else {
                    const int32_t lo = v > outBase ? v : outBase;
                    const int32_t hi = v + 16 < opV ? v + 16 : opV;
                    for (int32_t p = lo; p < hi; p++) {
                        outAligned[p] = outRing[p & (OUT_RING - 1)];
                    }
                }
            }
        }
        flushedV = opV & ~15;  // stays granule-aligned: a later flush re-stores the partial granule from the ring, whole
        order();
    }

    // n copies of one byte (Zstd RLE blocks)
    __device__ __forceinline__ void fill(int32_t op, uint32_t value, int32_t n)
    {
        enter();
        while (n > 0) {
            const int32_t c = n < CHUNK ? n : CHUNK;
            for (int32_t k = g; k < c; k += GS) {
                out_put(op + k, value);
            }
            op += c;
            n -= c;
            flush_complete(op);
        }
    }

    // literals: n input bytes at ip -> output at op (n arbitrary; input ring refilled, output flushed as we go)
    __device__ __forceinline__ void copy_literals(int32_t ip, int32_t op, int32_t n)
    {
        enter();
        if (UNIFIED && n > 4 * GS) {
            while (n > 0) {
                const int32_t c = n < CHUNK ? n : CHUNK;
                ensure_input(ip, c);
                copy_dwords<IN_RING>(inRing, ip + inBase, op + outBase, c);
                ip += c;
                op += c;
                n -= c;
                if (!PH_FLUSH || n > 0) {  // (deferred flushes: a run of one chunk leaves its flush to the next memory phase)
                    flush_complete(op);
                }
            }
            return;
        }
        if (n <= 4 * GS) {
            ensure_input(ip, n);
            copy_small<IN_RING>(inRing, ip + inBase, op + outBase, n);
            if (!PH_FLUSH) {
                flush_complete(op + n);
            }
            return;
        }
        while (n > 0) {
            const int32_t c = n < CHUNK ? n : CHUNK;
            ensure_input(ip, c);
            if (c <= (GS > 4 ? GS : 4)) {
                for (int32_t k = g; k < c; k += GS) {
                    out_put(op + k, in_u8(ip + k));
                }
            }
            else {
                copy_dwords<IN_RING>(inRing, ip + inBase, op + outBase, c);
            }
            ip += c;
            op += c;
            n -= c;
            flush_complete(op);
        }
    }

    // back-reference: out[op+k] = out[op-offset+k], byte-sequential semantics, n arbitrary.
    // Processed in chunks of <= CHUNK bytes.  Inside a chunk starting at c0 every byte j reads
    // c0 - offset + (j mod offset): for offset >= chunk length that is the plain source, for shorter
    // offsets the period is folded so all sources lie BEFORE the chunk (no intra-chunk dependency).
    __device__ __forceinline__ void copy_match(int32_t op, int32_t offset, int32_t n)
    {
        enter();
        if (UNIFIED && !(n <= 4 * GS && offset >= n)) {
            if (stage != nullptr) {
                // A trip never reads what it writes (c <= dist); a distance shorter than a chunk doubles once a whole
                // period has been written -- the data is periodic, so out[x] = out[x - 2 * dist] as well.
                int32_t dist = offset;
                while (n > 0) {
                    int32_t c = n < CHUNK ? n : CHUNK;
                    c = c < dist ? c : dist;
                    order();
                    if (dist <= LDS_REACH) {
                        copy_dwords<OUT_RING>(outRing, op + outBase - dist, op + outBase, c);
                    }
                    else {
                        // flushed long ago (dist > LDS_REACH >= 2 * CHUNK): 64 source bytes land in the staging area,
                        // then the same move as every other copy.  Reading past c stays inside this block's output.
                        if (PH_FLUSH && op + outBase - dist + CHUNK > flushedV) {  // (deferred flushes: what is read back must have left)
                            flush_complete(op);
                            order();
                        }
                        *(u32x4*)(stage + 16 * g) = ld16(outAligned + outBase + (op - dist) + 16 * g);
                        order();
                        copy_dwords<CHUNK>(stage, 0, op + outBase, c);
                    }
                    op += c;
                    n -= c;
                    if (dist < CHUNK) {
                        dist += dist;
                    }
                    if (!PH_FLUSH || n > 0) {
                        flush_complete(op);
                    }
                }
                return;
            }
        }
        if (n <= 4 * GS && offset >= n) {
            order();
            if (offset <= LDS_REACH) {
                copy_small<OUT_RING>(outRing, op + outBase - offset, op + outBase, n);
            }
            else {
                if (PH_FLUSH && op + outBase - offset + 4 * GS > flushedV) {
                    flush_complete(op);
                    order();
                }
                put4(op + outBase + 4 * g, ld4(outAligned + outBase + (op - offset) + 4 * g), n - 4 * g);  // flushed long ago (see below)
            }
            if (!PH_FLUSH) {
                flush_complete(op + n);
            }
            return;
        }
        int32_t c0 = op;
        while (n > 0) {
            const int32_t c = n < CHUNK ? n : CHUNK;
            order();
            if (offset <= LDS_REACH) {
                if (offset >= c && c <= (GS > 4 ? GS : 4)) {
                    for (int32_t j = g; j < c; j += GS) {
                        out_put(c0 + j, out_get(c0 - offset + j));
                    }
                }
                else if (offset >= c) {
                    copy_dwords<OUT_RING>(outRing, c0 + outBase - offset, c0 + outBase, c);
                }
                else {
                    // ceil(2^32 / offset): j / offset == umulhi(j, inv) exactly for j < CHUNK (offset == 1: quotient is j)
                    const uint32_t inv = offset == 1 ? 0u : (0xFFFFFFFFu / (uint32_t)offset + 1u);
                    for (int32_t j = g; j < c; j += GS) {
                        const uint32_t q = offset == 1 ? (uint32_t)j : __umulhi((uint32_t)j, inv);
                        out_put(c0 + j, out_get(c0 - offset + (j - (int32_t)q * offset)));
                    }
                }
            }
            else {
                // far: offset > LDS_REACH >= 2*CHUNK, so the sources of this chunk were flushed to HBM at least CHUNK bytes ago
                const uint8_t* src = outAligned + outBase + (c0 - offset);
                if (c <= (GS > 4 ? GS : 4)) {
                    for (int32_t j = g; j < c; j += GS) {
                        out_put(c0 + j, src[j]);
                    }
                }
                else {
                    const int32_t dV = c0 + outBase;
                    const int32_t head = (4 - (dV & 3)) & 3;
                    const int32_t nd = (c - head) >> 2;
                    const int32_t t0 = head + 4 * nd;
                    for (int32_t k = g; k < head; k += GS) {
                        outRing[(dV + k) & (OUT_RING - 1)] = src[k];
                    }
                    for (int32_t j = g; j < nd; j += GS) {
                        *(uint32_t*)(outRing + ((dV + head + 4 * j) & (OUT_RING - 1))) = ld4(src + head + 4 * j);
                    }
                    for (int32_t k = t0 + g; k < c; k += GS) {
                        outRing[(dV + k) & (OUT_RING - 1)] = src[k];
                    }
                }
            }
            c0 += c;
            n -= c;
            flush_complete(c0);
        }
    }
};

}  // namespace achip
