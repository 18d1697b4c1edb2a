// tools/hostemu/emu.cpp -- runs the lane-private decoder kernels on the CPU (sequential lanes are exact when a kernel
// uses no cross-lane operation: the GS=1 instantiations of the ring decoders and the lane-per-block decoders with an LDS window).
#if !defined(HOSTEMU_NO_RINGS_LOCKSTEP)
#define HOSTEMU_RINGS_LOCKSTEP 1  // achip_rings.h: the lanes of a group meet where the device's lockstep makes them meet
#endif
#include "hip/hip_runtime.h"
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern "C" { long long achip_emu_counters[16]; }  // development counters of kernels under emulation (ACHIP_EMU_COUNT)
#include "../../aircompressor_amd/csrc/lz4_decompress_v2.hip"
#include "../../aircompressor_amd/csrc/snappy_decompress_v2.hip"
#include "../../aircompressor_amd/csrc/lz4_decompress_v7.hip"
#include "../../aircompressor_amd/csrc/snappy_decompress_v5.hip"
#include "../../aircompressor_amd/csrc/hadoop_streams.hip"
#include "../../aircompressor_amd/csrc/lz4_frame.hip"
#include "../../aircompressor_amd/csrc/snappy_frame.hip"
#include <vector>
// the probes of the decoders' auto mode are not built here: the probe statistics stay
// zero, which picks the ring decoders
namespace achip {
hipError_t launch_snappy_element_sample(const BatchArgs&, hipStream_t, int32_t*, int32_t, int32_t) { return hipSuccess; }
hipError_t launch_lz4_sequence_sample(const BatchArgs&, hipStream_t, int32_t*, int32_t, int32_t) { return hipSuccess; }
hipError_t launch_lz4_mixed_groups(const BatchArgs&, hipStream_t, int32_t*, int32_t) { return hipSuccess; }
}  // namespace achip
extern "C" int emu_batch(int op, const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff,
                         const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 0};
    if (op == 24 || op == 25 || op == 26 || op == 27 || op == 34 || op == 35 || op == 36 || op == 37) {  // two-pass decoders (24 / 25 LZ4 with a lane per block parsing, 26 / 27 with a wavefront per block; 34 / 35 and 36 / 37 Snappy likewise); odd ops: a tiny arena, so that blocks fall back
        const bool snappy = op >= 34, tiny = (op & 1) != 0;
        achip::g_lz4_parse_mode = op == 26 || op == 27 ? 2 : 1;
        achip::g_snappy_parse_mode = op == 36 || op == 37 ? 2 : 1;
        static std::vector<uint8_t> scratch;
        const int64_t bytes = tiny ? 4096 + ((n * 12 + 4095) & ~4095LL) + 4 * 4096 : achip::lz4_twopass_scratch_bytes(n);
        scratch.assign((size_t)bytes, 0xCD);
        a.ringPad = 16;
        return snappy ? achip::launch_snappy_decompress_twopass(a, nullptr, scratch.data(), bytes, 1, 0, 2, nullptr)
                      : achip::launch_lz4_decompress_twopass(a, nullptr, scratch.data(), bytes, 1, 0, 2, nullptr);
    }
    if (op == 16 || op == 17) {  // default ring decoders at GS = 1 (compact / large rings)
        a.ringPad = 16;
        return achip::launch_lz4_decompress_rings(a, nullptr, 1, op - 16, nullptr);
    }
    if (op == 12 || op == 13) {
        a.ringPad = 16;
        return achip::launch_snappy_decompress_rings(a, nullptr, 1, op - 12, nullptr);
    }
    // the ring decoders with a lane GROUP per block (the product's shape: 4 lanes; 16 and 64 for large blocks): 44 / 46 / 48 LZ4, 54 / 56 / 58
    // Snappy; the product's ring padding (80 bytes: the staging area of far matches included)
    if (op == 44 || op == 46 || op == 48 || op == 54 || op == 56 || op == 58) {
        a.ringPad = 80;
        const int gs = (op % 10) == 4 ? 4 : ((op % 10) == 6 ? 16 : 64);
        return op < 50 ? achip::launch_lz4_decompress_rings(a, nullptr, gs, 0, nullptr) : achip::launch_snappy_decompress_rings(a, nullptr, gs, 0, nullptr);
    }
    if (op == 49 || op == 59) {  // the latency class (ring class 3): a wavefront and 128 KiB of LDS history per block
        a.ringPad = 80;
        return op == 49 ? achip::launch_lz4_decompress_rings(a, nullptr, 4, 3, nullptr) : achip::launch_snappy_decompress_rings(a, nullptr, 4, 3, nullptr);
    }
    return -1;
}

// Hadoop block streams (hadoop_streams.hip): op 0 = decompress, 1 = compress; the cooperative writer / reader kernels under the fiber emulator
extern "C" int emu_hadoop(int op, int snappy, int bufferSize, int variant, const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase,
                          const int64_t* dstOff, const int32_t* dstCap, int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 16};
    static std::vector<uint8_t> scratch;
    const int64_t bytes = op == 0 ? achip::hadoop_decompress_scratch_bytes(n, bufferSize) : achip::hadoop_compress_scratch_bytes(n);
    scratch.assign((size_t)bytes, 0xCD);
    static std::vector<uint8_t> auxBuffer;
    const achip::AuxScratch aux{[](void*, int64_t bytes) -> void* { auxBuffer.assign((size_t)bytes, 0xCD); return auxBuffer.data(); }, nullptr};
    return op == 0 ? achip::launch_hadoop_decompress(a, nullptr, scratch.data(), snappy != 0, bufferSize, variant, &aux) : achip::launch_hadoop_compress(a, nullptr, scratch.data(), snappy != 0, bufferSize);
}

// the executor for records of any length (achip_seqexec2.h exec_records, used by the Zstd pipeline): one block, one wavefront
namespace {
void emu_exec_records_kernel(const uint64_t* rec, int32_t n, const uint8_t* lit, int32_t litSize, uint8_t* out, int32_t outLimit, int32_t* result)
{
    __shared__ __attribute__((aligned(16))) uint8_t win[achip::sx2::WIN_DEFAULT + 16];
    achip::sx2::RecordSource S{rec, n};
    bool bad = false;
    const int32_t produced = achip::sx2::exec_records<>(win, S, lit, litSize, out, outLimit, (int)threadIdx.x, bad);
    if (threadIdx.x == 0) {
        result[0] = produced;
        result[1] = bad ? 1 : 0;
    }
}
}  // namespace
extern "C" int emu_exec_records(const uint64_t* rec, int32_t n, const uint8_t* lit, int32_t litSize, uint8_t* out, int32_t outLimit, int32_t* result)
{
    hipLaunchKernelGGL(emu_exec_records_kernel, dim3(1), dim3(64), 0, nullptr, rec, n, lit, litSize, out, outLimit, result);
    return 0;
}

// LZ4 frames (lz4_frame.hip), reader variant 1: walk, the frames' blocks through the two-pass decoder, fold (the wavefront-per-item kernel
// behind it moves bytes between lanes in hardware order and is not emulated: irregular items come back with whatever it made of them)
extern "C" int emu_lz4frame(int variant, const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff, const int32_t* dstCap,
                            int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 16};
    static std::vector<uint8_t> scratch, auxBuffer;
    scratch.assign((size_t)achip::lz4frame_decompress_scratch_bytes(n, variant), 0xCD);
    const achip::AuxScratch aux{[](void*, int64_t bytes) -> void* { auxBuffer.assign((size_t)bytes, 0xCD); return auxBuffer.data(); }, nullptr};
    return achip::launch_lz4frame_decompress(a, nullptr, scratch.data(), variant, &aux);
}

// x-snappy-framed streams (snappy_frame.hip), reader variant 2: walk, the chunks through the two-pass Snappy decoder, CRC verification, fold
extern "C" int emu_snappyframed(int variant, const uint8_t* srcBase, const int64_t* srcOff, const int32_t* srcLen, uint8_t* dstBase, const int64_t* dstOff, const int32_t* dstCap,
                                int32_t* outLen, int32_t* status, int64_t* errOffset, int32_t n)
{
    achip::BatchArgs a{srcBase, srcOff, srcLen, dstBase, dstOff, dstCap, outLen, status, errOffset, n, 16};
    static std::vector<uint8_t> scratch, auxBuffer;
    scratch.assign((size_t)achip::snappyframed_decompress_scratch_bytes(n), 0xCD);
    const achip::AuxScratch aux{[](void*, int64_t bytes) -> void* { auxBuffer.assign((size_t)bytes, 0xCD); return auxBuffer.data(); }, nullptr};
    return achip::launch_snappyframed_decompress(a, nullptr, scratch.data(), variant, &aux);
}
