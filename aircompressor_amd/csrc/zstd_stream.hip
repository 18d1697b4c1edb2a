// zstd_stream.hip -- the Zstd STREAM writer for gfx950: what ZstdOutputStream (M/zstd/ZstdOutputStream.java:30-221) puts on
// its sink for one write(buffer, 0, n) + close() -- the way the reference's stream harness drives it (T/HadoopCodecCompressor.java:57-72).
// The encoder is zstd_compress_body.h (the frame compressor's device code); this translation unit adds the stream's driver and kernel, and
// is a unit of its own so that zstd_compress.hip compiles to exactly the code it was verified as.
//
// What differs from the frame compressor (ZstdFrameCompressor.compress):
//   :48-58    the parameters are those for an UNKNOWN size -- CompressionParameters.compute(3, -1) returns the default row as it stands
//             (:259-261): window 2^20, chain 2^16, hash 2^17 whatever n is;
//   :107-131  the buffer becomes min(2 n, 4 MiB) (at least a block) and only a FULL 4 MiB buffer is flushed before close(): below 4 MiB the
//             stream is one chunk whose size the frame header announces (the bytes are then the frame compressor's for those parameters);
//   :154-221  from 4 MiB on: a header without the content size; a flush writes whole blocks and keeps window + one block; then the tables and
//             the buffer slide (BlockCompressionState.java:35-49) -- here c.in moves and every position is relative to it -- while
//             c.windowBaseOffset stays where enforceMaxDistance put it, as in the reference (the blocks behind a slide find no match until
//             the position has caught up with it).  That part was written without a GPU at hand; it is byte-identical with the CPU
//             restatement of the Java writer (the test suite's checker) under tools/hostemu's access-granular lockstep (check_enc.py --chunked: 4.5 MB and 6.4 MB streams, one and two
//             slides).  `chunked` == 0 (context option zstd.stream.chunked) refuses such streams instead.
#include "zstd_compress_body.h"

namespace achip {

namespace zc {
constexpr int32_t STREAM_MAX_BUFFER = 4 << 20;
// a wavefront's block buffer lives where the frame compressor's match kernel keeps a wavefront's tables (zstd_compress_scratch_bytes: at least as many of
// those as slabs); its stride is theirs
constexpr int64_t STREAM_BLOCK_BUFFER_BYTES = (int64_t)4 * (HASH_TABLE_INTS + CHAIN_TABLE_INTS);

__device__ int32_t zstd_stream_item(Ctx& c, Shared& sh, int chunked, uint8_t* blockBuf)
{
    const int32_t n = c.inLen;
    const int32_t outputLimit = c.outCap;
    if (n >= STREAM_MAX_BUFFER && (chunked == 0 || n >= (1 << 30))) {
        c.failStatus = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
        return -1;
    }
    c.searchLength = LEVEL3[0][3];
    c.windowLog = LEVEL3[0][0];
    c.windowSize = 1 << c.windowLog;
    c.blockSize = MAX_BLOCK_SIZE;
    c.chainLog = LEVEL3[0][1];
    c.hashLog = LEVEL3[0][2];
    const uint8_t* const whole = c.in;
    const int32_t windowSize = c.windowSize, blockMax = c.blockSize;
    int32_t bufferLength = 2 * n < 4 * windowSize ? 2 * n : 4 * windowSize;  // growBufferIfNecessary :107-120 (n < 2^30)
    bufferLength = bufferLength > blockMax ? bufferLength : blockMax;
    int32_t output = 0;
    ZC_CHECK(c, outputLimit - output >= 4);  // writeMagic (ZstdFrameCompressor.java:55-61)
    st4(c.out + output, 0xFD2FB528u);
    output += 4;
    // a fresh CompressionContext (:50)
    c.offset0 = 1;
    c.offset1 = 4;
    c.tempOffset0 = c.tempOffset1 = 0;
    c.windowBaseOffset = 0;
    wave_fill((uint8_t*)c.hashTable, 0, 4 << c.hashLog, c.lane);
    wave_fill((uint8_t*)c.chainTable, 0, 4 << c.chainLog, c.lane);
    for (int i = c.lane; i < 256; i += 64) {
        sh.huf[0].numberOfBits[i] = 0;
        sh.huf[1].numberOfBits[i] = 0;
    }
    sh.huf[0].maxSymbol = 0;
    sh.huf[0].maxNumberOfBits = 0;
    sh.huf[1].maxSymbol = 0;
    sh.huf[1].maxNumberOfBits = 0;
    c.previousTable = 0;
    c.temporaryTable = 1;
    c.previousCandidate = 0;
    c.temporaryCandidate = 1;
    __syncthreads();
    wave_mem_order();
    int32_t outputSize = 0;
    int32_t offset = 0, position = 0, length = n;  // uncompressedOffset, uncompressedPosition (relative to c.in), bytes not yet written
    bool first = true;
    for (int closing = 0; closing <= 1; closing++) {
        for (;;) {
            int32_t chunk;
            if (!closing) {
                if (length <= 0) {
                    break;
                }
                const int32_t writeSize = length < bufferLength - position ? length : bufferLength - position;  // write :93-104
                position += writeSize;
                length -= writeSize;
                if (!(bufferLength >= 4 * windowSize && position == bufferLength)) {
                    continue;  // compressIfNecessary :122-131 (the input is used up: close() follows)
                }
                chunk = ((position - offset - windowSize - blockMax) / blockMax) * blockMax;
            }
            else {
                chunk = position - offset;
            }
            if (first) {  // writeFrameHeader(inputSize = closing ? chunk : -1, windowSize) (ZstdFrameCompressor.java:64-121)
                first = false;
                const int32_t inputSize = closing ? chunk : -1;
                const int32_t contentSizeDescriptor = inputSize == -1 ? 0 : (inputSize >= 256 ? 1 : 0) + (inputSize >= 65536 + 256 ? 1 : 0);
                int32_t fhd = (contentSizeDescriptor << 6) | 0x04;
                const bool singleSegment = inputSize != -1 && windowSize >= inputSize;
                // (the stream composes the header in its own buffer and hands the sink exactly these bytes: the sink's room is checked against them)
                ZC_CHECK(c, outputLimit - output >= 1 + (singleSegment ? 0 : 1) + (contentSizeDescriptor == 0 ? (singleSegment ? 1 : 0) : (contentSizeDescriptor == 1 ? 2 : 4)));
                if (singleSegment) {
                    fhd |= 0x20;
                }
                c.out[output++] = (uint8_t)fhd;
                if (!singleSegment) {
                    c.out[output++] = (uint8_t)((c.windowLog - 10) << 3);
                }
                if (contentSizeDescriptor == 0) {
                    if (singleSegment) {
                        c.out[output++] = (uint8_t)inputSize;
                    }
                }
                else if (contentSizeDescriptor == 1) {
                    st2(c.out + output, (uint32_t)(inputSize - 256));
                    output += 2;
                }
                else {
                    st4(c.out + output, (uint32_t)inputSize);
                    output += 4;
                }
                outputSize = outputLimit - output;
            }
            do {  // :186-204
                const int32_t blockSize = chunk < blockMax ? chunk : blockMax;
                const bool lastBlock = closing != 0 && blockSize == chunk;
                // writeCompressedBlock (ZstdFrameCompressor.java:181-204) works in the STREAM's own buffer of a fixed length (ZstdOutputStream.java:55-58),
                // whatever the sink's room is: the block is compressed into this wavefront's block buffer with exactly that room, and the sink --
                // the caller's buffer -- is checked against the bytes it is then handed (with the caller's remaining capacity as the
                // room, capacities below the bound failed where the Java stream succeeds)
                const int32_t privateLength = (blockMax + 3) + ((blockMax + 3) >> 8) + 8;
                int32_t compressedSize = 0;
                if (blockSize > 0) {
                    uint8_t* const sink = c.out;
                    c.out = blockBuf;
                    compressedSize = compress_block(c, sh, offset, blockSize, 3, privateLength - 3);
                    c.out = sink;
                    ZC_PROPAGATE(compressedSize);
                }
                if (compressedSize == 0) {
                    ZC_CHECK(c, blockSize + 3 <= outputSize);
                    st_le(c.out + output, (uint32_t)((lastBlock ? 1 : 0) | (0 << 1) | (blockSize << 3)), 3);
                    wave_mem_order();
                    group_copy<64>(c.out + output + 3, c.in + offset, blockSize, c.lane);
                    wave_mem_order();
                    compressedSize = 3 + blockSize;
                }
                else {
                    ZC_CHECK(c, compressedSize + 3 <= outputSize);
                    wave_mem_order();
                    group_copy<64>(c.out + output + 3, blockBuf + 3, compressedSize, c.lane);
                    wave_mem_order();
                    st_le(c.out + output, (uint32_t)((lastBlock ? 1 : 0) | (2 << 1) | (compressedSize << 3)), 3);
                    compressedSize += 3;
                }
                offset += blockSize;
                chunk -= blockSize;
                output += compressedSize;
                outputSize -= compressedSize;
            } while (chunk > 0);
            if (closing) {
                break;
            }
            // slide :212-219
            const int32_t slide = offset - windowSize;
            wave_mem_order();
            for (int32_t i = c.lane; i < (1 << c.hashLog); i += 64) {
                const int32_t v = c.hashTable[i] - slide;
                c.hashTable[i] = v & ~(v >> 31);
            }
            for (int32_t i = c.lane; i < (1 << c.chainLog); i += 64) {
                const int32_t v = c.chainTable[i] - slide;
                c.chainTable[i] = v & ~(v >> 31);
            }
            wave_mem_order();
            c.in += slide;
            offset -= slide;
            position -= slide;
        }
    }
    ZC_CHECK(c, outputLimit - output >= 4);  // :206-211
    const uint64_t hash = wave_xxh64(whole, n, c.lane);
    st4(c.out + output, (uint32_t)hash);
    output += 4;
    return output;
}

}  // namespace zc

// the same persistent loop and per-wavefront slab as zstd_compress_kernel (zstd_compress.hip)
__global__ __launch_bounds__(64) void zstd_stream_kernel(BatchArgs a, uint8_t* slabs, uint8_t* blockBufs, int32_t* nextItem, int32_t count, int32_t chunked)
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
        const int32_t block = item;
        Ctx c;
        c.in = a.srcBase + a.srcOff[block];
        c.inLen = a.srcLen[block];
        c.out = a.dstBase + a.dstOff[block];
        c.outCap = a.dstCap[block];
        c.lane = lane;
        c.batchProbe = 2;  // the window match finder (zstd_dfast_mw.h)
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
        int32_t r;
        if (c.inLen < 0 || c.outCap < 0) {
            r = -1;
            c.failStatus = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_BAD_ARGUMENT);
        }
        else {
            r = zstd_stream_item(c, sh, chunked, blockBufs + (size_t)blockIdx.x * STREAM_BLOCK_BUFFER_BYTES);
        }
        if (lane == 0) {
            a.outLen[block] = r >= 0 ? r : 0;
            a.status[block] = r >= 0 ? 0 : c.failStatus;
            a.errOffset[block] = 0;
        }
    }
}

// scratch: zstd_compress_scratch_bytes(nBlocks) (the frame compressor's: an item counter and a slab per wavefront)
hipError_t launch_zstd_stream_compress(const BatchArgs& a, hipStream_t stream, void* scratch, int chunked)
{
    if (a.nBlocks <= 0) {
        return hipSuccess;
    }
    uint8_t* base = (uint8_t*)scratch;
    int32_t* counter = (int32_t*)base;
    hipError_t e = hipMemsetAsync(counter, 0, 64, stream);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(a.nBlocks < 256 * 8 ? a.nBlocks : 256 * 8);  // (ZC_MAX_WAVES of zstd_compress.hip: what the scratch holds slabs for)
    uint8_t* const blockBufs = base + 4096 + (int64_t)grid * zc::SLAB_BYTES;    // (the match kernel's table slabs in zstd_compress_scratch_bytes' layout)
    hipLaunchKernelGGL(zstd_stream_kernel, dim3(grid), dim3(64), 0, stream, a, base + 4096, blockBufs, counter + 8, a.nBlocks, chunked);
    return hipGetLastError();
}

}  // namespace achip
