// tools/hostemu/emu_serial.cpp -- the wavefront-per-item kernels under the fiber emulator: the LZ4 frame reader (lz4_frame.hip variant 0), the
// x-snappy-framed and Hadoop stream readers' wavefront-per-stream kernels (variant 0), the one-kernel Zstd decoder.  They run one item per
// wavefront in wave-uniform control flow over Rings<64, ...>: wave_mem_order() is a rendezvous of the wave here (HOSTEMU_ORDER_IS_RENDEZVOUS),
// the rings' lockstep points are rendezvous of the 64-lane group.  A library of its own: the other emulated kernels need wave_mem_order()
// to stay what it is on the device.
#define HOSTEMU_RINGS_LOCKSTEP 1
#define HOSTEMU_ORDER_IS_RENDEZVOUS 1
#include "hip/hip_runtime.h"
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern "C" { long long achip_emu_counters[16]; }
#include "../../aircompressor_amd/csrc/lz4_frame.hip"
#include "../../aircompressor_amd/csrc/snappy_frame.hip"
#include "../../aircompressor_amd/csrc/hadoop_streams.hip"
#include "../../aircompressor_amd/csrc/zstd_decompress.hip"
#include <vector>
namespace achip {
// the list paths' launchers and the Zstd pipeline (not part of this library: the wavefront-per-item kernels only)
hipError_t launch_lz4_decompress_twopass(const BatchArgs&, hipStream_t, void*, int64_t, int, int, int, const int32_t*) { return hipSuccess; }
hipError_t launch_snappy_decompress_twopass(const BatchArgs&, hipStream_t, void*, int64_t, int, int, int, const int32_t*) { return hipSuccess; }
int64_t twopass_scratch_bytes(int32_t, int64_t) { return 0; }
hipError_t launch_lz4_decompress_rings(const BatchArgs&, hipStream_t, int, int, const int32_t*) { return hipSuccess; }
int lz4_ring_group_for(int32_t) { return 4; }
int snappy_ring_group_for(int32_t) { return 4; }
hipError_t launch_lz4_sequence_sample(const BatchArgs&, hipStream_t, int32_t*, int32_t, int32_t) { return hipSuccess; }
hipError_t launch_snappy_decompress_rings(const BatchArgs&, hipStream_t, int, int, const int32_t*) { return hipSuccess; }
hipError_t launch_snappy_element_sample(const BatchArgs&, hipStream_t, int32_t*, int32_t, int32_t) { return hipSuccess; }
hipError_t launch_lz4_mixed_groups(const BatchArgs&, hipStream_t, int32_t*, int32_t) { return hipSuccess; }
int64_t zstd_decompress_pipe_scratch_bytes(int32_t, int32_t) { return 0; }
hipError_t launch_zstd_decompress_pipe(const BatchArgs&, hipStream_t, void*, void*, int32_t, const ZstdMbProvider*) { return hipSuccess; }
void* zstd_decompress_pipe_general_scratch(void* scratch, int32_t, int32_t) { return scratch; }
}  // namespace achip

extern "C" int emu_lz4frame_serial(const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff, const int32_t* dstCap,
                                   int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 16};
    static std::vector<uint8_t> scratch;
    scratch.assign(4096, 0);
    return achip::launch_lz4frame_decompress(a, nullptr, scratch.data(), 0, nullptr);
}

// op 8: x-snappy-framed streams, 10 / 12: Hadoop LZ4 / Snappy block streams, 4: Zstd frames -- each through its wavefront-per-item kernel
extern "C" int emu_serial(int op, int bufferSize, const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff,
                          const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 16};
    static std::vector<uint8_t> scratch;
    if (op == 6) {
        scratch.assign(4096, 0);
        return achip::launch_lz4frame_decompress(a, nullptr, scratch.data(), 0, nullptr);
    }
    if (op == 8) {
        scratch.assign((size_t)achip::snappyframed_decompress_scratch_bytes(n), 0xCD);
        return achip::launch_snappyframed_decompress(a, nullptr, scratch.data(), 0, nullptr);
    }
    if (op == 10 || op == 12) {
        scratch.assign((size_t)achip::hadoop_decompress_scratch_bytes(n, bufferSize), 0xCD);
        return achip::launch_hadoop_decompress(a, nullptr, scratch.data(), op == 12, bufferSize, 0, nullptr);
    }
    if (op == 4) {
        scratch.assign((size_t)achip::zstd_decompress_general_scratch_bytes() + 4096, 0xCD);
        return achip::launch_zstd_decompress(a, nullptr, scratch.data(), (int64_t)scratch.size(), 0, 0, nullptr);
    }
    return -1;
}
