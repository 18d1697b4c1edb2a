// tools/hostemu/emu_zstd.cpp -- the Zstd decode pipeline (zstd_decompress_pipe.hip: parse / literals / sequences / execute / checksum, and
// the multi-block stages) under the fiber emulator.  The sequence stage gives an item to a quad of lanes: its DPP broadcasts and LDS
// hand-overs are quad-level rendezvous here (hip/hip_runtime.h).  The one-kernel decoder that takes the pipeline's fallback list moves
// bytes between lanes in hardware order and is NOT emulated: items on the fallback list are reported to the caller instead.
#include "hip/hip_runtime.h"
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern "C" { long long achip_emu_counters[16]; }
#include "../../aircompressor_amd/csrc/zstd_decompress_pipe.hip"
#include <vector>
namespace achip {
namespace {
// the predefined FSE tables, built the way zstd_decompress.hip's zstd_default_tables_kernel builds them
void emu_default_tables_kernel(zd::FseTable* dflt)
{
    using namespace zd;
    __shared__ TableShared sh;
    Ctx c;
    c.in = nullptr;
    c.inLen = 0;
    c.out = nullptr;
    c.outCap = 0;
    c.lit = nullptr;
    c.R = nullptr;
    c.lane = threadIdx.x;
    c.detail = 0;
    c.errOff = 0;
    const int16_t* norms[3] = {LL_DEFAULT_NORM, OF_DEFAULT_NORM, ML_DEFAULT_NORM};
    const int32_t maxSym[3] = {35, 28, 52};
    const int32_t logs[3] = {6, 5, 6};
    for (int k = 0; k < 3; k++) {
        __syncthreads();
        for (int i = c.lane; i <= maxSym[k]; i += 64) {
            sh.norm[i] = norms[k][i];
        }
        __syncthreads();
        fse_build(c, sh, sh.fse[k], maxSym[k], logs[k], 0);
        __syncthreads();
        for (int i = c.lane; i < 512; i += 64) {
            dflt[k].e[i] = sh.fse[k].e[i];
        }
    }
}
std::vector<int32_t> g_fallback;
}  // namespace
hipError_t launch_zstd_decompress_prepare(hipStream_t, void* generalScratch, const zd::FseTable** dflt)
{
    zd::FseTable* t = (zd::FseTable*)((uint8_t*)generalScratch + 4096);
    hipLaunchKernelGGL(emu_default_tables_kernel, dim3(1), dim3(64), 0, nullptr, t);
    *dflt = t;
    return hipSuccess;
}
hipError_t launch_zstd_decompress_list(const BatchArgs&, hipStream_t, void*, const int32_t* list, const int32_t* listCount)
{
    g_fallback.assign(list, list + *listCount);
    return hipSuccess;
}
int64_t zstd_decompress_general_scratch_bytes() { return 4096 + (int64_t)sizeof(zd::FseTable) * 3 + 4096; }
}  // namespace achip

// runs the pipeline over the batch; items it handed to the fallback list get status -1000 and are listed in fallback[0 .. return value)
namespace {
std::vector<uint8_t> g_mbScratch;
int64_t g_mbMaxBytes = 0;  // > 0: larger requests are refused (the stages then ask for less)
void* emu_mb_get(void*, int64_t bytes)
{
    if (g_mbMaxBytes > 0 && bytes > g_mbMaxBytes) {
        return nullptr;
    }
    g_mbScratch.assign((size_t)bytes, 0xCD);
    return g_mbScratch.data();
}
}  // namespace

// a stream decoded a step at a time (launch_zstd_stream_step): the caller (check_zstd.py --stream) plays the host's part -- the walk over the
// block headers, the stand-in frame header, the history
extern "C" int64_t emu_zstd_stream_carry_bytes() { return achip::zstd_stream_carry_bytes(); }
extern "C" void emu_zstd_stream_carry_init(void* carry) { achip::zstd_stream_carry_init(carry); }
extern "C" int emu_zstd_stream_step(void* carry, const uint8_t* src, int32_t srcLen, int32_t blocks, uint8_t* out, int32_t startPos, int32_t outLimit, int32_t closing,
                                    int32_t hasChecksum, uint32_t expected, int32_t* result)
{
    static std::vector<uint8_t> scratch;
    const int64_t bytes = achip::zstd_stream_step_scratch_bytes(blocks);
    scratch.assign((size_t)bytes, 0xCD);
    return (int)achip::launch_zstd_stream_step(nullptr, scratch.data(), bytes, carry, src, srcLen, blocks, out, startPos, outLimit, closing, hasChecksum, expected, result);
}

// passBlocks: blocks per pass of the multi-block stages (0: multi-block frames go to the fallback list); counters: the pipeline's 64 counter words
extern "C" int emu_zstd_pipe(const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff, const int32_t* dstCap,
                             int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n, int32_t tile, int32_t execMode, int32_t* fallback, int32_t passBlocks,
                             int32_t* counters, int64_t mbMaxBytes)
{
    g_mbMaxBytes = mbMaxBytes;
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 16};
    static std::vector<uint8_t> scratch;
    const int64_t bytes = achip::zstd_decompress_pipe_scratch_bytes(n, tile);
    scratch.assign((size_t)bytes, 0xCD);
    achip::g_zstd_pipe_exec = execMode & 3;
    achip::g_zstd_seq_waves = (execMode >> 4) & 7 ? (execMode >> 4) & 7 : 1;  // (bits 4 .. 6: wavefronts per workgroup of the sequence stage, with full workgroups)
    achip::g_zstd_seq_spread = (execMode >> 4) & 7 ? 1 : 256;
    achip::g_zstd_lit_items = ((execMode >> 2) & 3) == 1 ? 8 : (((execMode >> 2) & 3) == 2 ? 10 : (((execMode >> 2) & 3) == 3 ? 13 : 16));  // (bits 2, 3: items per wavefront of the literal stage; 13: the split table layout)
    if ((execMode >> 7) & 1) achip::g_zstd_lit_items = 20;  // (bit 7: 16 items, symbols and lengths by symbol)
    achip::g_fallback.clear();
    for (int32_t i = 0; i < n; i++) {
        status[i] = -999;  // "not written"
    }
    achip::ZstdMbProvider mbp{emu_mb_get, nullptr, passBlocks};
    achip::launch_zstd_decompress_pipe(a, nullptr, scratch.data(), achip::zstd_decompress_pipe_general_scratch(scratch.data(), n, tile), tile, passBlocks > 0 ? &mbp : nullptr);
    memcpy(counters, scratch.data(), 256);
    for (size_t k = 0; k < achip::g_fallback.size(); k++) {
        fallback[k] = achip::g_fallback[k];
        status[achip::g_fallback[k]] = -1000;
    }
    return (int)achip::g_fallback.size();
}
