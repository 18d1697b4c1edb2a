#!/bin/bash
# builds tools/hostemu/libemu_all_asan.so (the decode kernels' sources + AddressSanitizer) and runs asan_fuzz.py under the ASan runtime:
#   tools/hostemu/run_asan_fuzz.sh <seed> <rounds> [families...]
#   tools/hostemu/run_asan_fuzz.sh --enc <seed> <rounds>        the encoders and writers (libemu_enc_asan.so: lockstep + ASan; asan_enc.py)
#   tools/hostemu/run_asan_fuzz.sh --ubsan <seed> <rounds> [families...]   the decoders under UBSan instead (signed overflow, shifts, bounds, division)
cd "$(dirname "$0")/../.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
if [ "$1" = "--ubsan" ]; then
    shift
    if [ ! -f tools/hostemu/libemu_all_ubsan.so ] || [ tools/hostemu/emu_all.cpp -nt tools/hostemu/libemu_all_ubsan.so ] || [ -n "$(find aircompressor_amd/csrc tools/hostemu/hip -newer tools/hostemu/libemu_all_ubsan.so -name '*.h*' | head -1)" ]; then
        $CLANG -O1 -g -std=c++17 -fPIC -shared -fsanitize=signed-integer-overflow,shift,bounds,integer-divide-by-zero -shared-libsan -fno-omit-frame-pointer \
            -I tools/hostemu -I include -I aircompressor_amd/csrc -o tools/hostemu/libemu_all_ubsan.so tools/hostemu/emu_all.cpp || exit 1
    fi
    HOSTEMU_LIB=libemu_all_ubsan.so LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.ubsan_standalone-x86_64.so) exec python tools/hostemu/asan_fuzz.py "$@"
fi
if [ "$1" = "--enc" ]; then
    shift
    if [ ! -f tools/hostemu/libemu_enc_asan.so ] || [ tools/hostemu/emu_enc.cpp -nt tools/hostemu/libemu_enc_asan.so ] || [ -n "$(find aircompressor_amd/csrc tools/hostemu/hip -newer tools/hostemu/libemu_enc_asan.so -name '*.h*' | head -1)" ]; then
        $CLANG -O2 -g -std=c++17 -fPIC -shared -Wl,-Bsymbolic-functions -fsanitize=address -shared-libasan -fno-omit-frame-pointer -fsanitize-coverage=inline-8bit-counters,trace-loads,trace-stores \
            -I tools/hostemu -I include -I aircompressor_amd/csrc -o tools/hostemu/libemu_enc_asan.so tools/hostemu/emu_enc.cpp || exit 1
    fi
    LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 exec python tools/hostemu/asan_enc.py "$@"
fi
if [ ! -f tools/hostemu/libemu_all_asan.so ] || [ tools/hostemu/emu_all.cpp -nt tools/hostemu/libemu_all_asan.so ] || [ -n "$(find aircompressor_amd/csrc tools/hostemu/hip -newer tools/hostemu/libemu_all_asan.so -name '*.h*' | head -1)" ]; then
    $CLANG -O1 -g -std=c++17 -fPIC -shared -fsanitize=address -shared-libasan -fno-omit-frame-pointer -I tools/hostemu -I include -I aircompressor_amd/csrc \
        -o tools/hostemu/libemu_all_asan.so tools/hostemu/emu_all.cpp || exit 1
fi
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 python tools/hostemu/asan_fuzz.py "$@"
