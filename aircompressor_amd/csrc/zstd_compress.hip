// zstd_compress.hip -- batched Zstd frame encode (level 3) for gfx950, bit-exact with the Java encoder.
//
// Replaces ZstdFrameCompressor.compress(level 3) and everything under it:
//   frame / block assembly ... M/zstd/ZstdFrameCompressor.java:52-432
//   parameters ............... M/zstd/CompressionParameters.java:36-145,256-324 (level 3 rows: always DFAST)
//   match finder ............. M/zstd/DoubleFastBlockCompressor.java:28-256 (+ BlockCompressionState, RepeatedOffsets)
//   sequence store / codes ... M/zstd/SequenceStore.java:20-160
//   sequence entropy coding .. M/zstd/SequenceEncoder.java:34-342, FseCompressionTable.java:18-155,
//                              FiniteStateEntropy.java:153-521 (optimalTableLog, normalizeCounts(2), writeNormalizedCounts, compress)
//   literal entropy coding ... M/zstd/HuffmanCompressionTable.java:27-437, HuffmanCompressor.java:26-135,
//                              HuffmanCompressionContext.java
//   bit output ............... M/zstd/BitOutputStream.java:20-90
//
// One wavefront per frame (batch item), persistent grid.  The parse (two hash tables, repcodes, lazy +1 probe) and
// the entropy-table construction are inherently serial and are executed wave-uniformly: every lane computes the same
// values and stores the same value to the same address, so there are no cross-lane hazards and no barriers in the
// serial code.  The lanes split the bulk work: table clears, 64 x 8-byte match-length counts, literal copies,
// code generation, histograms (LDS atomics), the four Huffman streams (one lane each, after their sizes have
// been computed with a wave reduction), the XXH64 stripes.
// Hash tables (up to 2^17 + 2^16 ints) and the sequence store live in a per-wave slab in HBM (L2-resident while the
// frame is being encoded); all entropy tables live in LDS.
#include "zstd_compress_body.h"

namespace achip {

namespace zc {
// per-item scratch of the two-kernel path: [record | seqOffset | seqLitLen | seqMatchLen | literals]
constexpr int64_t ITEM_BYTES = 256 + (int64_t)3 * 4 * MAX_SEQUENCES + MAX_BLOCK_SIZE + 64;
__device__ __forceinline__ void point_item_scratch(Ctx& c, uint8_t* itemScratch)
{
    uint8_t* p = itemScratch + 256;
    c.seqOffset = (int32_t*)p;
    p += 4 * MAX_SEQUENCES;
    c.seqLitLen = (int32_t*)p;
    p += 4 * MAX_SEQUENCES;
    c.seqMatchLen = (int32_t*)p;
    p += 4 * MAX_SEQUENCES;
    c.litBuf = p;
}
}  // namespace zc

// K_m: the match finder alone (DoubleFastBlockCompressor.compressBlock) for the eligible items [first, first + count)
__global__ __launch_bounds__(64) ACHIP_WAVES_PER_EU(4, 8) void zstd_match_kernel(BatchArgs a, uint8_t* tableSlabs, uint8_t* itemScratch, int32_t first, int32_t count, int32_t* nextItem, int32_t batchProbe)
{
    using namespace zc;
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    uint8_t* slab = tableSlabs + (size_t)blockIdx.x * (4 * (HASH_TABLE_INTS + CHAIN_TABLE_INTS));
    for (;;) {
        __syncthreads();
        if (lane == 0) {
            item = atomicAdd(nextItem, 1);
        }
        __syncthreads();
        const int32_t slot = item;
        if (slot >= count) {
            return;
        }
        const int32_t block = first + slot;
        uint8_t* mine = itemScratch + (size_t)slot * ITEM_BYTES;
        int32_t* rec = (int32_t*)mine;
        Ctx c;
        c.in = a.srcBase + a.srcOff[block];
        c.inLen = a.srcLen[block];
        c.out = nullptr;
        c.outCap = 0;
        c.lane = lane;
        c.batchProbe = batchProbe;
        c.failStatus = 0;
        c.pre = nullptr;
        if (!split_eligible(c.inLen)) {
            if (lane == 0) {
                rec[0] = 0;
            }
            continue;
        }
        c.hashTable = (int32_t*)slab;
        c.chainTable = (int32_t*)(slab + 4 * HASH_TABLE_INTS);
        point_item_scratch(c, mine);
        c.codeLL = c.codeML = c.codeOF = nullptr;
        compute_parameters(c);
        c.offset0 = 1;  // a fresh CompressionContext (ZstdFrameCompressor.java:162)
        c.offset1 = 4;
        c.tempOffset0 = c.tempOffset1 = 0;
        c.windowBaseOffset = 0;
        wave_fill((uint8_t*)c.hashTable, 0, 4 << c.hashLog, lane);
        wave_fill((uint8_t*)c.chainTable, 0, 4 << c.chainLog, lane);
        wave_mem_order();
        c.literalsLength = 0;
        c.sequenceCount = 0;
        c.longLengthField = 0;
        c.longLengthPosition = 0;
        const int32_t lastLiteralsSize = match_finder(c, 0, c.inLen);
        wave_mem_order();
        group_copy<64>(c.litBuf + c.literalsLength, c.in + c.inLen - lastLiteralsSize, lastLiteralsSize, lane);
        c.literalsLength += lastLiteralsSize;
        if (lane == 0) {
            rec[1] = c.sequenceCount;
            rec[2] = c.literalsLength;
            rec[3] = c.longLengthField;
            rec[4] = c.longLengthPosition;
            rec[5] = c.tempOffset0;
            rec[6] = c.tempOffset1;
            rec[0] = 1;
        }
    }
}

__global__ __launch_bounds__(64) void zstd_compress_kernel(BatchArgs a, uint8_t* slabs, int32_t* nextItem, uint8_t* itemScratch, int32_t first, int32_t count)
{
    using namespace zc;
    __shared__ Shared sh;
    __shared__ int32_t item;
    const int lane = threadIdx.x;
    // the three predefined tables (SequenceEncoder.java:66-68), once per wave
    fse_initialize(sh, sh.dflt[0], LL_DEFAULT_NORM, 35, 6);
    fse_initialize(sh, sh.dflt[1], OF_DEFAULT_NORM, 28, 5);
    fse_initialize(sh, sh.dflt[2], ML_DEFAULT_NORM, 52, 6);
    uint8_t* slab = slabs + (size_t)blockIdx.x * SLAB_BYTES;
    for (;;) {
        __syncthreads();
        if (lane == 0) {
            item = atomicAdd(nextItem, 1);
        }
        __syncthreads();
        if (item >= count) {
            return;
        }
        const int32_t block = first + item;
        Ctx c;
        c.in = a.srcBase + a.srcOff[block];
        c.inLen = a.srcLen[block];
        c.out = a.dstBase + a.dstOff[block];
        c.outCap = a.dstCap[block];
        c.lane = lane;
        c.batchProbe = a.ringPad == 1 ? 0 : (a.ringPad == 3 ? 2 : 1);  // variant 1 = serial probing, 3 = many matches per window
        c.failStatus = 0;
        c.pre = nullptr;
        uint8_t* p = slab;
        c.hashTable = (int32_t*)p;
        p += 4 * HASH_TABLE_INTS;
        c.chainTable = (int32_t*)p;
        p += 4 * CHAIN_TABLE_INTS;
        c.seqOffset = (int32_t*)p;
        p += 4 * MAX_SEQUENCES;
        c.seqLitLen = (int32_t*)p;
        p += 4 * MAX_SEQUENCES;
        c.seqMatchLen = (int32_t*)p;
        p += 4 * MAX_SEQUENCES;
        c.codeLL = p;
        p += MAX_SEQUENCES;
        c.codeML = p;
        p += MAX_SEQUENCES;
        c.codeOF = p;
        p += MAX_SEQUENCES;
        c.litBuf = p;
        if (itemScratch != nullptr) {
            uint8_t* mine = itemScratch + (size_t)item * ITEM_BYTES;
            if (((const int32_t*)mine)[0] == 1) {  // the match kernel has been here: sequence store and literals live in the item's scratch
                c.pre = (const int32_t*)mine;
                point_item_scratch(c, mine);
            }
        }
        int32_t r;
        if (c.inLen < 0 || c.outCap < 0) {
            r = -1;
            c.failStatus = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
        }
        else {
            r = zstd_compress_item(c, sh);
        }
        if (lane == 0) {
            a.outLen[block] = r >= 0 ? r : 0;
            a.status[block] = r >= 0 ? 0 : c.failStatus;
            a.errOffset[block] = 0;
        }
    }
}

namespace {
constexpr int ZC_MAX_WAVES = 256 * 8;     // entropy (and one-kernel) path: 154 VGPRs => 2 waves per SIMD
constexpr int ZM_MAX_WAVES = 256 * 24;    // match kernel: 62 VGPRs, no LDS; 16 -> 24 waves per CU measured +4..12 %, 28 no better
constexpr int32_t ZC_TILE = 8192;         // items per pass of the two-kernel path (their scratch: 608 KB each)
constexpr int64_t ZM_TABLE_BYTES = 4 * (zc::HASH_TABLE_INTS + zc::CHAIN_TABLE_INTS);
}

int64_t zstd_compress_scratch_bytes(int32_t nBlocks)
{
    // three regions, each sized for the waves / items a batch of nBlocks can put in flight (a one-frame call needs 2.7 MB)
    const int64_t n = nBlocks < 1 ? 1 : nBlocks;
    const int64_t tile = n < ZC_TILE ? n : ZC_TILE;
    const int64_t cWaves = n < ZC_MAX_WAVES ? n : ZC_MAX_WAVES, mWaves = n < ZM_MAX_WAVES ? n : ZM_MAX_WAVES;
    return 4096 + cWaves * zc::SLAB_BYTES + mWaves * ZM_TABLE_BYTES + tile * zc::ITEM_BYTES;
}

// variant 0 (default): match-finder kernel + entropy kernel for one-block inputs, one kernel for the rest; 1: the same with
// serial probing; 3: the same with the window match finder (zstd_dfast_mw.h); 2: everything in the one kernel; 100: timing aid (one kernel, stop after the match finder)
hipError_t launch_zstd_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int64_t scratchBytes, int variant)
{
    (void)scratchBytes;
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* base = (uint8_t*)scratch;
    int32_t* counter = (int32_t*)base;
    uint8_t* slabs = base + 4096;
    uint8_t* tableSlabs = slabs + (int64_t)(a.nBlocks < ZC_MAX_WAVES ? a.nBlocks : ZC_MAX_WAVES) * zc::SLAB_BYTES;
    uint8_t* itemScratch = tableSlabs + (int64_t)(a.nBlocks < ZM_MAX_WAVES ? a.nBlocks : ZM_MAX_WAVES) * ZM_TABLE_BYTES;
    const bool split = variant == 0 || variant == 1 || variant == 3;
    const int32_t tile = split ? ZC_TILE : a.nBlocks;
    for (int32_t first = 0; first < a.nBlocks; first += tile) {
        const int32_t count = a.nBlocks - first < tile ? a.nBlocks - first : tile;
        hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
        if (e != hipSuccess) return e;
        if (split) {
            const unsigned mgrid = (unsigned)(count < ZM_MAX_WAVES ? count : ZM_MAX_WAVES);
            hipLaunchKernelGGL(zstd_match_kernel, dim3(mgrid), dim3(64), 0, stream, a, tableSlabs, itemScratch, first, count, counter, variant == 1 ? 0 : (variant == 3 ? 2 : 1));
        }
        const unsigned grid = (unsigned)(count < ZC_MAX_WAVES ? count : ZC_MAX_WAVES);
        hipLaunchKernelGGL(zstd_compress_kernel, dim3(grid), dim3(64), 0, stream, a, slabs, counter + 8, split ? itemScratch : (uint8_t*)nullptr, first, count);
    }
    return hipGetLastError();
}

}  // namespace achip
