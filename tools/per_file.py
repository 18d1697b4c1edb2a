"""DEBUG: LZ4 decode rate of every 64 KiB slice of the committed corpus sample, one batch of identical blocks per slice,
for two decoder variants (which data favours which decoder)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import aircompressor_amd as A

dev = torch.device("cuda", 0)
codec = A.HipBatchCodec(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sample = np.fromfile(os.path.join(root, "tests", "golden", "corpus_sample.bin"), dtype=np.uint8)
meta = json.load(open(os.path.join(root, "tests", "golden", "corpus_sample.json")))
names = ["%s+%d" % (m["file"], m["offset"]) for m in meta]
bs, n = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 32768
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 7]
i64 = dict(dtype=torch.int64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
max_c = codec.lib.achip_lz4_max_compressed_length(bs)
cs = (max_c + 15) // 16 * 16
mixes = [[int(x) for x in m.split("+")] for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else [[k] for k in range(sample.size // bs)]
for mix in mixes:  # a batch cycles over the slices of its mix (one slice: identical blocks)
    k = mix[0]
    plain1 = torch.cat([torch.from_numpy(sample[j * bs:(j + 1) * bs].copy()) for j in mix]).to(dev)
    assert n % len(mix) == 0
    plain = plain1.repeat(n // len(mix))
    so = torch.arange(n, **i64) * bs; sl = torch.full((n,), bs, **i32)
    comp = torch.empty(n * cs + 64, dtype=torch.uint8, device=dev); co = torch.arange(n, **i64) * cs; cc = torch.full((n,), max_c, **i32)
    cl = torch.zeros(n, **i32); st = torch.zeros(n, **i32); eo = torch.zeros(n, **i64)
    torch.cuda.synchronize()
    codec.launch(A.OP_LZ4_COMPRESS, plain, so, sl, comp, co, cc, cl, st, eo, n); codec.synchronize()
    out = torch.empty(n * bs + 64, dtype=torch.uint8, device=dev)
    ol = torch.zeros(n, **i32)
    row = "%-14s %-28s ratio %5.2f" % ("+".join(map(str, mix)), names[k] if len(mix) == 1 else "(mix)", n * bs / float(cl.sum()))
    for v in variants:
        codec.native.set_option("lz4.decompress.variant", v)
        f = lambda: codec.launch(A.OP_LZ4_DECOMPRESS, comp, co, cl, out, so, sl, ol, st, eo, n)
        f(); codec.synchronize()
        e0, e1 = codec.event(), codec.event()
        codec.record(e0); f(); f(); codec.record(e1); codec.synchronize()
        t = codec.elapsed_ms(e0, e1) / 2
        ok = bool((out[:n * bs].view(n // len(mix), len(mix) * bs) == plain1.unsqueeze(0)).all()) and int((st != 0).sum()) == 0
        row += "  v%d %7.1f GiB/s%s" % (v, n * bs / t / 1e-3 / 2**30, "" if ok else " WRONG")
    print(row, flush=True)
