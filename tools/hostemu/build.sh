#!/bin/bash
# Rebuilds the emulator libraries the check_*.py scripts load (the same commands tests/test_host_logic.py runs): tools/hostemu/build.sh [zstd] [enc] [all] [serial] [emu]
# The scripts load whatever library lies there -- after a change to a kernel source, build first.
cd "$(dirname "$0")/../.."
CLANG=$(command -v clang++ || echo /opt/rocm/lib/llvm/bin/clang++)
I="-I tools/hostemu -I include -I aircompressor_amd/csrc"
for t in ${@:-zstd enc all serial emu}; do
  case $t in
    zstd)   $CLANG -O1 -std=c++17 -fPIC -shared $I -o tools/hostemu/libemu_zstd.so tools/hostemu/emu_zstd.cpp ;;
    enc)    $CLANG -O2 -std=c++17 -fPIC -shared -fno-omit-frame-pointer -fsanitize-coverage=inline-8bit-counters,trace-loads,trace-stores $I -o tools/hostemu/libemu_enc.so tools/hostemu/emu_enc.cpp ;;
    all)    $CLANG -O1 -std=c++17 -fPIC -shared $I -o tools/hostemu/libemu_all.so tools/hostemu/emu_all.cpp ;;
    serial) $CLANG -O1 -std=c++17 -fPIC -shared $I -o tools/hostemu/libemu_serial.so tools/hostemu/emu_serial.cpp ;;
    emu)    $CLANG -O1 -std=c++17 -fPIC -shared $I -o tools/hostemu/libemu.so tools/hostemu/emu.cpp ;;
  esac || exit 1
  echo "built $t"
done
