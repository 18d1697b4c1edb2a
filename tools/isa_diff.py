#!/usr/bin/env python3
"""Which kernels compile to different instructions than at <commit>?   tools/isa_diff.py <commit> [file.hip ...]

Compiles aircompressor_amd/csrc/*.hip (or the named files) for gfx950 at <commit> and in the working tree (hipcc -S, device only, no
GPU needed) and compares every function's instruction stream (labels and comments normalised).  Use: after a refactor that is meant
to leave a kernel alone -- a template parameter with a default, code moved to a header, a new caller in the same translation unit --
this says whether it did; a kernel whose instructions are identical to a build that passed the GPU tests needs no new GPU run.
(Round 2 learnt it the hard way: a second caller in the same unit changed what the inliner did to the Zstd encoder's kernel.)"""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def functions(path):
    out = {}
    txt = open(path).read()
    for m in re.finditer(r"\.type\s+(\S+),@function\n(.*?)\.Lfunc_end\d+:", txt, re.S):
        lines = [l.strip() for l in m.group(2).splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
        lines = [re.sub(r"\.LBB\d+_\d+", "L", re.sub(r"\s*;.*$", "", l)) for l in lines]
        out[m.group(1)] = lines[1:]
    return out


def build(tree, src, dst):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-I", os.path.join(tree, "include"), "-x", "hip", "-S",
                    "--cuda-device-only", "-o", dst, src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    commit = sys.argv[1]
    only = [os.path.basename(f) for f in sys.argv[2:]]
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        tar = subprocess.run(["git", "-C", ROOT, "archive", commit, "aircompressor_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", old], input=tar, check=True)
        jobs = []
        for tree, tag in ((old, "old"), (ROOT, "new")):
            for src in sorted(glob.glob(os.path.join(tree, "aircompressor_amd", "csrc", "*.hip"))):
                if only and os.path.basename(src) not in only:
                    continue
                jobs.append((tree, src, os.path.join(tmp, "%s-%s.s" % (tag, os.path.basename(src)[:-4]))))
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda j: build(*j), jobs))
        before = {}
        for p in glob.glob(os.path.join(tmp, "old-*.s")):
            before.update(functions(p))
        for p in sorted(glob.glob(os.path.join(tmp, "new-*.s"))):
            now = functions(p)
            changed = [k for k, v in now.items() if k in before and before[k] != v]
            added = [k for k in now if k not in before]
            print("%-26s %3d identical, %d changed, %d new" % (os.path.basename(p)[4:-2], len(now) - len(changed) - len(added), len(changed), len(added)))
            for k in changed:
                print("    changed: " + k)
            for k in added:
                print("    new:     " + k)


if __name__ == "__main__":
    main()
