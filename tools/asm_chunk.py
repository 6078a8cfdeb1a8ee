"""VALU / LDS / SALU instruction counts of the SSV kernel's chunk loop on its usual path (no separators, no matrix edge,
four-step windows), per region: the chunk's prologue, its eight windows, its epilogue.  python tools/asm_chunk.py [file.s] [-v]"""
import collections
import sys

path = next((a for a in sys.argv[1:] if a.endswith(".s")), "havac_dev-hip-amdgcn-amd-amdhsa-gfx950.s")
verbose = "-v" in sys.argv
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN5havac15ssv_diag_kernel")][0]
body = lines[start:]
w0 = [i for i, l in enumerate(body) if "expand_for_windowILi0E" in l and l.startswith(".LBB")][0]
sw = [i for i, l in enumerate(body) if "step_windowsIJ" in l and l.startswith(".LBB")][0]
hdr = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and i < w0][-1]
end = [i for i, l in enumerate(body) if i > sw and "s_branch" in l or (i > sw and "s_cbranch" in l and "LBB" in l)][0:40]


def count(lo, hi):
    c = collections.Counter()
    skip = False
    last = ""
    for l in body[lo:hi]:
        t = l.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            skip = False
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        if skip:
            continue
        c[op] += 1
        if verbose:
            print("   ", t[:100])
        # the block behind a forward branch over it is a rare path: special entries, the middle test of unsafe chunks
        if op.startswith("s_cbranch") and not t.split()[-1].startswith(".LBB7_1") is None:
            skip = True
        last = op
    return c


# the loop ends at the back edge: first s_branch/s_cbranch to the header label after the windows
hname = body[hdr].split(":")[0]
back = [i for i, l in enumerate(body) if i > sw and hname in l and ("s_branch" in l or "s_cbranch" in l)]
stop = back[0] + 1 if back else sw + 120
for name, (lo, hi) in dict(prologue=(hdr, w0), windows=(w0, sw), epilogue=(sw, stop)).items():
    if verbose:
        print("==", name)
    c = count(lo, hi)
    v = sum(n for k, n in c.items() if k.startswith("v_"))
    print(name, "VALU", v, "LDS", sum(n for k, n in c.items() if k.startswith("ds_")), "SALU", sum(n for k, n in c.items() if k.startswith("s_")),
          "VMEM", sum(n for k, n in c.items() if k.startswith(("global_", "buffer_"))))
    print("   ", {k: n for k, n in c.most_common(40) if k.startswith("v_")})
