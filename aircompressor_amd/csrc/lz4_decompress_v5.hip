// lz4_decompress_v5.hip -- batched LZ4 block decode for gfx950: a lane per block, copies straight between global buffers.
//
// Same contract and the same Java-order checks as lz4_decompress_v2.hip (M/lz4/Lz4RawDecompressor.java:35-198).
//
// The ring decoders (v2) spend their time issuing instructions: a wavefront walks 16 blocks, one sequence of each at a
// time, and executes the union of the paths those 16 state machines take (profiles/r01_notes.md: 2.3x the instructions
// of a converged wavefront; ~128 SIMD cycles per sequence on text).  Here every LANE owns a block (64 blocks per
// wavefront) and a trip of the loop is the same straight line for all of them:
//   parse    token, length extensions, offset -- from the lane's LDS window on its compressed stream (16-byte granules,
//            requested ahead; literal bytes are jumped over) -- with the Java checks in their order;
//   copy     ONE wavefront-wide copy step moves the literal run and the match of all 64 sequences: all loads of the step
//            are issued before its stores, so a trip costs about one memory round trip.  Up to 32 bytes of a copy are
//            moved by its own lane (two overlapping 16-byte pieces, or 8/4/2/1); what is longer is cut into 16-byte
//            chunks that are dealt out evenly to the 64 lanes (consecutive lanes take consecutive chunks), so neither a
//            64 KiB literal run nor a skewed mix of lengths leaves lanes idle.  Copies are exact.
//   A match that overlaps itself (offset < length) or reads this trip's literals takes extra steps: one period first,
//   then -- the written part repeating the period -- twice as much per step.
// There are no output rings: sources are read from, and bytes written to, the output buffer itself (the L2 holds the
// recent window; a wavefront's vector memory operations are performed in program order).
#include "achip_lanecopy.h"

namespace achip {

template <int IN_DW, bool NT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void lz4_decompress_lanecopy_kernel(BatchArgs a, const int32_t* mixedGroups)
{
    using namespace sp;
    if (mixedGroups != nullptr && lz4_pick(mixedGroups, batch_count(a)) != LZ4_PICK_LANECOPY) {  // auto mode: another decoder takes this batch
        return;
    }
    __shared__ uint32_t ldsIn[IN_DW * 64];
    __shared__ CopyScratch S;
    const int lane = threadIdx.x;
    const int64_t block = (int64_t)blockIdx.x * 64 + lane;
    const bool have = block < batch_count(a);
    const uint8_t* in = have ? a.srcBase + a.srcOff[block] : a.srcBase;
    uint8_t* out = have ? a.dstBase + a.dstOff[block] : a.dstBase;
    const int32_t inLimit = have ? a.srcLen[block] : 0;
    const int32_t outLimit = have ? a.dstCap[block] : 0;

    LaneInput<IN_DW> R;
    R.init(ldsIn + lane, in, inLimit);

    int32_t st = 0;
    int32_t eo = 0;
    int32_t ip = 0;
    int32_t op = 0;
    bool done = !have;
    const int32_t fastOutLimit = outLimit - 8;

#define LZ4_FAIL(detail, off)                          \
    {                                                  \
        st = mk_status(ACHIP_CLASS_MALFORMED, detail); \
        eo = (int32_t)(off);                           \
        done = true;                                   \
    }

    if (have) {
        if (inLimit == 0) {  // :48-50
            st = mk_status(ACHIP_CLASS_MALFORMED, ACHIP_D_LZ4_INPUT_EMPTY);
            done = true;
        }
        else if (outLimit == 0) {  // :52-57 (the Java method returns -1 here)
            if (!(inLimit == 1 && in[0] == 0)) {
                st = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_LZ4_EMPTY_OUTPUT);
            }
            done = true;
        }
    }

    // the copies in progress, as offsets: literal run (litRem bytes left: in[litPos..] -> out[litOut..]) and match (rem
    // bytes left at out[cur..], distance dist -- the offset, doubled with every whole period written)
    int32_t litPos = 0, litOut = 0, litRem = 0;
    int32_t cur = 0, rem = 0, dist = 0;
    int32_t periodic = 0;  // out[periodic .. cur) repeats with the match's offset: dist may be any multiple of it that stays inside
    const uint8_t* const inEnd = in + inLimit;
    const uint8_t* const outEnd = out + outLimit;
    int32_t tokenMl = 0;     // low nibble of the token whose match header is still to be parsed
    bool headerDue = false;  // the literal run was too long for the window: its match header is parsed when the run is copied
    while (__ballot(!done || rem > 0 || litRem > 0) != 0) {
        // ---- parse (lanes whose copies are complete): ONE 16-byte window at ip holds the token, a literal run of up to 12
        // bytes, the offset and the first match-length extension byte -- a whole sequence of the common kind.  A longer run
        // is copied from the input buffer over the next trips and its match header is parsed after it. ----
        HeadRegs h0;
        h0.A = u32x4{0, 0, 0, 0};
        h0.B = h0.A;
        bool have0 = false;
        if (rem == 0 && litRem == 0 && !done) {
            if (!headerDue && ip >= inLimit) {  // the Java loop condition :59
                done = true;
            }
            else {
                R.ensure_input(ip, 20);
                const u32x4 W = R.in_u128(ip);
                uint32_t hdr = W.x;  // header bytes: offset (2), first extension byte
                bool parseHeader = headerDue;
                if (!headerDue) {
                    const int32_t token = (int32_t)(W.x & 0xFF);
                    tokenMl = token & 0xF;
                    ip++;
                    int32_t lit = token >> 4;  // :62-77
                    if (lit == 0xF) {
                        if (ip >= inLimit) {
                            LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                        }
                        else {
                            int32_t v = (int32_t)((W.x >> 8) & 0xFF);  // first extension byte: in the window
                            ip++;
                            lit += v;
                            while (v == 255 && ip < inLimit - 15) {
                                R.ensure_input(ip, 4);
                                v = (int32_t)R.in_u8(ip++);
                                lit = (int32_t)((uint32_t)lit + (uint32_t)v);
                            }
                        }
                    }
                    if (!done && lit < 0) {
                        LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                    }
                    bool lastLiterals = false;
                    if (!done) {
                        const int64_t litEnd = (int64_t)ip + lit;
                        const int64_t litOutLimit = (int64_t)op + lit;
                        if (litOutLimit > fastOutLimit - 4 || litEnd > inLimit - 8) {  // :82-96 last literals
                            if (litOutLimit > outLimit) {
                                LZ4_FAIL(ACHIP_D_LZ4_LAST_LITERAL_OUTSIDE, ip);
                            }
                            else if (litEnd != inLimit) {
                                LZ4_FAIL(ACHIP_D_LZ4_INPUT_NOT_CONSUMED, ip);
                            }
                            else {
                                lastLiterals = true;
                            }
                        }
                    }
                    if (!done) {
                        litPos = ip;
                        litOut = op;
                        litRem = lit;
                        ip += lit;
                        op += lit;
                        cur = op;
                        if (lastLiterals) {
                            done = true;
                        }
                        if (lit <= 12) {  // (then there was no extension byte) the run sits in window bytes 1..12, the header behind it
                            h0.A = u32x4{alignbyte_u32(W.y, W.x, 1), alignbyte_u32(W.z, W.y, 1), alignbyte_u32(W.w, W.z, 1), W.w >> 8};
                            h0.B = h0.A;
                            have0 = lit > 0;
                            const uint32_t at = (uint32_t)lit + 1u;  // 1..13
                            const uint32_t lo = at < 4 ? W.x : (at < 8 ? W.y : (at < 12 ? W.z : W.w));
                            const uint32_t hi = at < 4 ? W.y : (at < 8 ? W.z : W.w);  // (at >= 12: bytes 12..15 are all in W.w, hi unused)
                            hdr = at >= 12 ? (W.w >> (8 * (at - 12))) : (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (at & 3)));
                            parseHeader = !lastLiterals;
                        }
                        else {
                            headerDue = !lastLiterals;
                        }
                    }
                }
                if (parseHeader && !done) {
                    headerDue = false;
                    const int32_t offset = (int32_t)(hdr & 0xFFFF);  // :113-119
                    ip += 2;
                    int32_t ml = 0;
                    if (offset == 0 || offset > op) {
                        LZ4_FAIL(ACHIP_D_LZ4_OFFSET_OUTSIDE, ip);
                    }
                    else {
                        ml = tokenMl;  // :122-138
                        bool bad = false;
                        if (ml == 0xF) {
                            if (ip > inLimit - 5) {
                                bad = true;
                            }
                            else {
                                int32_t v = (int32_t)((hdr >> 16) & 0xFF);  // first extension byte: in the window
                                ip++;
                                ml += v;
                                while (v == 255) {
                                    if (ip > inLimit - 5) {
                                        bad = true;
                                        break;
                                    }
                                    R.ensure_input(ip, 4);
                                    v = (int32_t)R.in_u8(ip++);
                                    ml = (int32_t)((uint32_t)ml + (uint32_t)v);
                                }
                            }
                        }
                        ml = (int32_t)((uint32_t)ml + 4u);
                        if (bad || ml < 0) {
                            LZ4_FAIL(ACHIP_D_LZ4_MALFORMED, ip);
                            ml = 0;
                        }
                        else {
                            const int64_t matchOutLimit = (int64_t)op + ml;
                            if (matchOutLimit > fastOutLimit - 4 && matchOutLimit > outLimit - 5) {  // :168-171
                                LZ4_FAIL(ACHIP_D_LZ4_LAST_5_LITERALS, ip);
                                ml = 0;
                            }
                        }
                    }
                    rem = ml;
                    dist = offset;
                    periodic = cur - offset;
                    op += ml;
                }
            }
        }
        // ---- copy.  Per trip a lane moves at most HEAD bytes of its literal run and HEAD bytes of its match itself and keeps
        // the rest for its next trips (it then parses nothing new) -- unless more than LONG bytes are left: those go out at once,
        // dealt to all lanes.  Of the match at most one period, and nothing while its source reaches into bytes of this trip.
        const int32_t n0 = (litRem > LONG || litRem < HEAD) ? litRem : HEAD;
        int32_t n1 = rem < dist ? rem : dist;
        n1 = (n1 > LONG || n1 < HEAD) ? n1 : HEAD;
        n1 = (litRem > n0 || dist < n0 + n1) ? 0 : n1;
        copy_step<true, NT>(S, lane, h0, have0, out + litOut, in + litPos, n0, inEnd, out + cur, out + cur - dist, n1, outEnd);
        litOut += n0;
        litPos += n0;
        litRem -= n0;
        cur += n1;
        rem -= n1;
        if (rem > dist && 2 * (int64_t)dist <= (int64_t)(cur - periodic)) {
            dist += dist;  // enough periods are written: out[x] = out[x - 2 * dist] holds as well
        }
    }
#undef LZ4_FAIL
    if (have) {
        a.outLen[block] = st == 0 ? op : 0;
        a.status[block] = st;
        a.errOffset[block] = (int64_t)eo;
    }
}

// counts the groups of 16 consecutive blocks whose compressed sizes differ by more than 2x
__global__ __launch_bounds__(256) void lz4_mixed_groups_kernel(BatchArgs a, int32_t* mixedGroups, int32_t minBlocks)
{
    const int32_t n = batch_count(a);
    if (n < minBlocks || n <= 0) {
        return;  // too few blocks for the lane-per-block decoder: the count stays 0
    }
    const int64_t group = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t first = group * 16;
    bool mixed = false;
    if (first < n) {
        int32_t lo = 0x7FFFFFFF, hi = 0;
        for (int64_t b = first; b < first + 16 && b < n; b++) {
            const int32_t len = a.srcLen[b];
            lo = len < lo ? len : lo;
            hi = len > hi ? len : hi;
        }
        mixed = (int64_t)hi > 2 * (int64_t)lo;
    }
    const int found = __popcll(__ballot(mixed));
    if ((threadIdx.x & 63) == 0 && found > 0) {
        atomicAdd(mixedGroups, found);
    }
}

// auto mode, LZ4 only: how long are the sequences?  1024 sampled blocks, the sequences in the first SAMPLE_HEAD bytes of each (at most 96; a
// lane per sample; headers only).  The head of the block is staged in LDS first: parsing it straight from global memory was one dependent
// round trip per header byte -- 0.11 .. 0.14 ms in front of every decode call, 1.5 % of the headline's step.
constexpr int SAMPLE_HEAD = 768;
constexpr int SAMPLE_STRIDE = SAMPLE_HEAD + 4;  // (an odd number of dwords: the lanes' heads start in different banks)

// the first min(inLimit, SAMPLE_HEAD) bytes of a lane's block into its LDS row (16-byte loads that stay inside the stream)
__device__ __forceinline__ int32_t sample_stage_head(uint8_t* h, const uint8_t* __restrict__ in, int32_t inLimit)
{
    const int32_t limit = inLimit < SAMPLE_HEAD ? inLimit : SAMPLE_HEAD;
#pragma unroll 4
    for (int32_t p = 0; p < limit; p += 16) {
        if (p + 16 <= inLimit) {
            const u32x4 v = ld16(in + p);
            __builtin_memcpy(h + p, &v, 16);
        }
        else {
            for (int32_t i = p; i < limit; i++) {
                h[i] = in[i];
            }
        }
    }
    return limit;
}

__global__ __launch_bounds__(64) void lz4_sequence_sample_kernel(BatchArgs a, int32_t* stats, int32_t minBlocks)
{
    const int32_t n = batch_count(a);
    if (n < minBlocks || n <= 0) {
        return;
    }
    __shared__ __attribute__((aligned(16))) uint8_t heads[64 * SAMPLE_STRIDE];
    const int32_t t = blockIdx.x * 64 + threadIdx.x;
    const int64_t block = (int64_t)t * n / 1024;
    uint8_t* const h = heads + threadIdx.x * SAMPLE_STRIDE;
    const int32_t inLimit = sample_stage_head(h, a.srcBase + a.srcOff[block], a.srcLen[block]);
    int64_t ip = 0;  // 64-bit: a run of length-extension bytes must not wrap the cursor
    int32_t seqs = 0;
    int64_t bytes = 0;
    while (ip < inLimit && seqs < 96) {
        const int32_t token = h[ip++];
        int64_t lit = token >> 4;
        if (lit == 15) {
            int32_t v = 255;
            while (v == 255 && ip < inLimit) {
                v = h[ip++];
                lit += v;
            }
        }
        bytes += lit;
        ip += lit;
        seqs++;
        if (ip + 2 > inLimit) {
            break;  // last literals, the end of the head (or nonsense: the decoders will say)
        }
        ip += 2;
        int64_t ml = token & 15;
        if (ml == 15) {
            int32_t v = 255;
            while (v == 255 && ip < inLimit) {
                v = h[ip++];
                ml += v;
            }
        }
        bytes += ml + 4;
    }
    bytes = bytes < 0 || bytes > (1 << 24) ? (1 << 24) : bytes;
    atomicAdd(stats + 1, seqs);
    atomicAdd(stats + 2, (int32_t)(bytes >> 2));  // (in units of 4 bytes: 1024 samples x 16 MiB stay inside 32 bits)
}

hipError_t launch_lz4_sequence_sample(const BatchArgs& a, hipStream_t stream, int32_t* stats, int32_t minBlocks)
{
    hipLaunchKernelGGL(lz4_sequence_sample_kernel, dim3(16), dim3(64), 0, stream, a, stats, minBlocks);
    return hipGetLastError();
}

hipError_t launch_lz4_mixed_groups(const BatchArgs& a, hipStream_t stream, int32_t* mixedGroups, int32_t minBlocks)
{
    const hipError_t e = hipMemsetAsync(mixedGroups, 0, 4 * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(((a.nBlocks + 15) / 16 + 255) / 256);
    hipLaunchKernelGGL(lz4_mixed_groups_kernel, dim3(grid), dim3(256), 0, stream, a, mixedGroups, minBlocks);
    return hipGetLastError();
}

hipError_t launch_lz4_decompress_lanecopy(const BatchArgs& a, hipStream_t stream, const int32_t* mixedGroups)
{
    const unsigned grid = (unsigned)((a.nBlocks + 63) / 64);
    hipLaunchKernelGGL((lz4_decompress_lanecopy_kernel<16, false>), dim3(grid), dim3(64), 0, stream, a, mixedGroups);
    return hipGetLastError();
}

}  // namespace achip
