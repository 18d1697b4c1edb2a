// tools/hostemu/emu_all.cpp -- every decode kernel in ONE emulator library, so that whole product paths run on the CPU: a container reader's
// list path AND the wavefront-per-item kernel behind it, the Zstd pipeline AND the one-kernel decoder that takes its fallback list.  What
// makes the mix possible is the SOFT order point (HOSTEMU_ORDER_IS_SOFT): wave_mem_order() pauses a lane until no lane of its wave can run
// further -- a barrier for the kernels that run a wavefront in uniform control flow, a harmless pause for lanes that go their own ways.
// Slower than libemu.so / libemu_zstd.so (every order point is a fiber switch): used by check_all.py, not by the quick checks.
#define HOSTEMU_ORDER_IS_SOFT 1
#include "emu.cpp"
#include "../../aircompressor_amd/csrc/zstd_decompress.hip"
#include "../../aircompressor_amd/csrc/zstd_decompress_pipe.hip"

namespace {
std::vector<uint8_t> g_allMb;
void* all_mb_get(void*, int64_t bytes)
{
    g_allMb.assign((size_t)bytes, 0xCD);
    return g_allMb.data();
}
}  // namespace

// the product's Zstd decode as achip_abi.cpp launches it: pipeline (multi-block stages included), then the one-kernel decoder over the fallback list
extern "C" int emu_zstd_full(const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff, const int32_t* dstCap,
                             int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n, int32_t variant, int32_t passBlocks, int32_t* counters)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 80};
    static std::vector<uint8_t> scratch;
    scratch.assign((size_t)achip::zstd_decompress_scratch_bytes(n, 0), 0xCD);
    const achip::ZstdMbProvider mbp{all_mb_get, nullptr, passBlocks};
    const int r = achip::launch_zstd_decompress(a, nullptr, scratch.data(), (int64_t)scratch.size(), variant, 0, passBlocks >= 16 ? &mbp : nullptr);
    if (counters != nullptr) {
        memcpy(counters, scratch.data(), 256);
    }
    return r;
}
