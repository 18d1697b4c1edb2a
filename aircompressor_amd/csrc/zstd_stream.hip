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
#include "achip_xxhash.h"

namespace achip {

namespace zc {
constexpr int32_t STREAM_MAX_BUFFER = 4 << 20;
// a wavefront's block buffer lives where the frame compressor's match kernel keeps a wavefront's tables (zstd_compress_scratch_bytes: at least as many of
// those as slabs); its stride is theirs
constexpr int64_t STREAM_BLOCK_BUFFER_BYTES = (int64_t)4 * (HASH_TABLE_INTS + CHAIN_TABLE_INTS);

// writeFrameHeader(inputSize, windowSize) (ZstdFrameCompressor.java:64-121) behind the magic: returns the new output position, or -1
__device__ int32_t stream_write_header(Ctx& c, int32_t output, int32_t outputLimit, int32_t inputSize)
{
    const int32_t windowSize = c.windowSize;
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
    return output;
}

// writeChunk's block loop :186-204: the blocks of `chunk` bytes from `offset` on (closing: the last one is marked last); 0, or -1
__device__ int32_t stream_write_chunk(Ctx& c, Shared& sh, uint8_t* blockBuf, int32_t& offset, int32_t chunk, bool closing, int32_t& output, int32_t& outputSize)
{
    const int32_t blockMax = c.blockSize;
    do {
        const int32_t blockSize = chunk < blockMax ? chunk : blockMax;
        const bool lastBlock = closing && blockSize == chunk;
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
    return 0;
}

// CompressionContext.slideWindow (BlockCompressionState.java:35-49): every table entry moves down by the slide, entries below it become 0
__device__ void stream_slide_tables(Ctx& c, int32_t slide)
{
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
}

// the parameters of an unknown size (:48-58)
__device__ __forceinline__ void stream_parameters(Ctx& c)
{
    c.searchLength = LEVEL3[0][3];
    c.windowLog = LEVEL3[0][0];
    c.windowSize = 1 << c.windowLog;
    c.blockSize = MAX_BLOCK_SIZE;
    c.chainLog = LEVEL3[0][1];
    c.hashLog = LEVEL3[0][2];
}

// a stream's fresh CompressionContext (:50)
__device__ void stream_fresh_context(Ctx& c, Shared& sh)
{
    stream_parameters(c);
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
}

__device__ int32_t zstd_stream_item(Ctx& c, Shared& sh, int chunked, uint8_t* blockBuf)
{
    const int32_t n = c.inLen;
    const int32_t outputLimit = c.outCap;
    if (n >= STREAM_MAX_BUFFER && (chunked == 0 || n >= (1 << 30))) {
        c.failStatus = mk_status(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
        return -1;
    }
    stream_fresh_context(c, sh);  // (the magic goes out first in the Java stream: nothing below depends on the order)
    const uint8_t* const whole = c.in;
    const int32_t windowSize = c.windowSize, blockMax = c.blockSize;
    int32_t bufferLength = 2 * n < 4 * windowSize ? 2 * n : 4 * windowSize;  // growBufferIfNecessary :107-120 (n < 2^30)
    bufferLength = bufferLength > blockMax ? bufferLength : blockMax;
    int32_t output = 0;
    ZC_CHECK(c, outputLimit - output >= 4);  // writeMagic (ZstdFrameCompressor.java:55-61)
    st4(c.out + output, 0xFD2FB528u);
    output += 4;
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
                output = stream_write_header(c, output, outputLimit, closing ? chunk : -1);
                ZC_PROPAGATE(output);
                outputSize = outputLimit - output;
            }
            ZC_PROPAGATE(stream_write_chunk(c, sh, blockBuf, offset, chunk, closing != 0, output, outputSize));
            if (closing) {
                break;
            }
            // slide :212-219
            const int32_t slide = offset - windowSize;
            stream_slide_tables(c, slide);
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

// ---- the same writer, a chunk per launch (achip_zstdstream_compress_begin / _feed / _finish: a stream of any length in the 4 MiB the Java stream
// buffers).  The host (achip_abi.cpp) keeps ZstdOutputStream's buffer on the device -- write() appends, a full buffer is flushed (compressIfNecessary
// :122-131), close() writes what is left -- and launches one step per writeChunk (:154-221); what a CompressionContext carries from chunk to chunk
// lives in this record between the launches (the match finder's tables stay where they are, in the stream's slab): the repeat offsets, the window's
// base, the Huffman tables a treeless block may reuse, the running XXH64 of the input. ----
struct ZstdOStreamState {
    int32_t offset0, offset1, windowBaseOffset;
    int32_t previousTable, temporaryTable, previousCandidate, temporaryCandidate;
    int32_t started;   // the context is set up, magic and frame header are out
    int32_t outSize;   // (out) bytes the step wrote
    int32_t status;    // (out) 0, or the step's failure
    int32_t pad[2];
    Xxh64Stream hash;
    zc::HufCTable huf[2];
};
int64_t zstd_ostream_state_bytes() { return (int64_t)sizeof(ZstdOStreamState); }
int64_t zstd_ostream_slab_bytes() { return zc::SLAB_BYTES + zc::STREAM_BLOCK_BUFFER_BYTES; }

// buf: the stream's buffer (position 0 = its first byte: the window in front of `offset`); the chunk is [offset, offset + chunk)
__global__ __launch_bounds__(64) void zstd_ostream_step_kernel(ZstdOStreamState* st, uint8_t* slab, const uint8_t* buf, int32_t offset, int32_t chunk, int32_t closing, uint8_t* out,
                                                               int32_t outCap)
{
    using namespace zc;
    __shared__ Shared sh;
    const int lane = threadIdx.x;
    fse_initialize(sh, sh.dflt[0], LL_DEFAULT_NORM, 35, 6);
    fse_initialize(sh, sh.dflt[1], OF_DEFAULT_NORM, 28, 5);
    fse_initialize(sh, sh.dflt[2], ML_DEFAULT_NORM, 52, 6);
    Ctx c;
    c.in = buf;
    c.inLen = offset + chunk;
    c.out = out;
    c.outCap = outCap;
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
    uint8_t* const blockBuf = slab + SLAB_BYTES;
    int32_t output = 0;
    int32_t result = 0;
    if (st->started == 0) {  // (uniform) the first chunk: writeMagic + writeFrameHeader (:167-178); the size is known only when it is also the last
        stream_fresh_context(c, sh);
        xxh64_stream_reset(&st->hash, lane);
        if (outCap < 4) {
            c.failStatus = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_ZSTD_MAX_OUTPUT);
            result = -1;
        }
        else {
            st4(c.out, 0xFD2FB528u);
            output = stream_write_header(c, 4, outCap, closing ? chunk : -1);
            result = output < 0 ? -1 : 0;
        }
    }
    else {
        stream_parameters(c);
        c.offset0 = st->offset0;
        c.offset1 = st->offset1;
        c.tempOffset0 = c.tempOffset1 = 0;
        c.windowBaseOffset = st->windowBaseOffset;
        c.previousTable = st->previousTable;
        c.temporaryTable = st->temporaryTable;
        c.previousCandidate = st->previousCandidate;
        c.temporaryCandidate = st->temporaryCandidate;
        for (int t = 0; t < 2; t++) {
            for (int i = lane; i < 256; i += 64) {
                sh.huf[t].values[i] = st->huf[t].values[i];
                sh.huf[t].numberOfBits[i] = st->huf[t].numberOfBits[i];
            }
            if (lane == 0) {
                sh.huf[t].maxSymbol = st->huf[t].maxSymbol;
                sh.huf[t].maxNumberOfBits = st->huf[t].maxNumberOfBits;
            }
        }
        __syncthreads();
    }
    if (result == 0) {
        xxh64_stream_update(&st->hash, buf + offset, chunk, lane);  // partialHash.update(uncompressed, uncompressedOffset, chunkSize) :181
        int32_t at = offset, outputSize = outCap - output;
        result = stream_write_chunk(c, sh, blockBuf, at, chunk, closing != 0, output, outputSize);
        if (result == 0 && closing) {  // :206-211
            if (outputSize < 4) {
                c.failStatus = mk_status(ACHIP_CLASS_OUTPUT_TOO_SMALL, ACHIP_D_ZSTD_MAX_OUTPUT);
                result = -1;
            }
            else {
                st4(c.out + output, (uint32_t)xxh64_stream_digest(&st->hash));
                output += 4;
            }
        }
        else if (result == 0) {
            stream_slide_tables(c, at - c.windowSize);  // :212-219 (the host moves the buffer)
        }
    }
    __syncthreads();
    for (int t = 0; t < 2; t++) {
        for (int i = lane; i < 256; i += 64) {
            st->huf[t].values[i] = sh.huf[t].values[i];
            st->huf[t].numberOfBits[i] = sh.huf[t].numberOfBits[i];
        }
    }
    if (lane == 0) {
        for (int t = 0; t < 2; t++) {
            st->huf[t].maxSymbol = sh.huf[t].maxSymbol;
            st->huf[t].maxNumberOfBits = sh.huf[t].maxNumberOfBits;
        }
        st->offset0 = c.offset0;
        st->offset1 = c.offset1;
        st->windowBaseOffset = c.windowBaseOffset;
        st->previousTable = c.previousTable;
        st->temporaryTable = c.temporaryTable;
        st->previousCandidate = c.previousCandidate;
        st->temporaryCandidate = c.temporaryCandidate;
        st->started = 1;
        st->outSize = result == 0 ? output : 0;
        st->status = result == 0 ? 0 : c.failStatus;
    }
}

hipError_t launch_zstd_ostream_step(hipStream_t stream, void* state, void* slab, const uint8_t* buf, int32_t offset, int32_t chunk, int32_t closing, uint8_t* out, int32_t outCap)
{
    hipLaunchKernelGGL(zstd_ostream_step_kernel, dim3(1), dim3(64), 0, stream, (ZstdOStreamState*)state, (uint8_t*)slab, buf, offset, chunk, closing, out, outCap);
    return hipGetLastError();
}

}  // namespace achip
