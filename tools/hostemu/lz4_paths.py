"""Which way the sequences of the LZ4 window encoder go (a counting build of the emulator library: -DACHIP_HOST_STATS).
    clang++ ... -DACHIP_HOST_STATS -o tools/hostemu/libemu_enc_stats.so tools/hostemu/emu_enc.cpp;  python tools/hostemu/lz4_paths.py"""
import ctypes, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools/hostemu")
import numpy as np
import check_enc
from check_enc import EncBatch, o
lib = ctypes.CDLL(os.path.join("tools", "hostemu", "libemu_enc_stats.so"))
check_enc.lib = lib
from tests import common
stats = (ctypes.c_longlong * 32).in_dll(lib, "g_zc_stats")
for name, data, _ in common.corpus_sample():
    for i in range(32): stats[i] = 0
    b = data[:65536]
    outs, status, _ = EncBatch(4).run(1, [b], [len(b) + len(b) // 255 + 16])
    ok = outs[0] == o.compress("lz4", b)
    print("%-28s ok=%s seq %5d fast %5d beyond %4d in-window cand %5d no-facts %4d mode2 %4d windows %5d" % (name, ok, stats[20], stats[21], stats[22], stats[23], stats[24], stats[25], stats[26]))
