"""Where the SSV kernel's scratch instructions are, relative to its packed adds (the hot loop), and an opcode count of a
line range.   python tools/asm_hot.py [file.s] [lo hi]"""
import collections
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".s") else "havac_dev-hip-amdgcn-amd-amdhsa-gfx950.s"
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN5havac15ssv_diag_kernel")][0]
end = [i for i, l in enumerate(lines) if i > start and ".Lfunc_end" in l][0]
body = lines[start:end]
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
if len(nums) == 2:
    c = collections.Counter()
    for l in body[nums[0]:nums[1]]:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        c[l.split()[0]] += 1
    print("VALU", sum(v for k, v in c.items() if k.startswith("v_")), "total", sum(c.values()))
    print(dict(c.most_common(30)))
    sys.exit(0)
b = collections.defaultdict(lambda: [0, 0, 0, 0])
for i, l in enumerate(body):
    k = i // 250
    if "v_pk_add_i16" in l:
        b[k][0] += 1
    if "scratch_store" in l:
        b[k][1] += 1
    if "scratch_load" in l:
        b[k][2] += 1
    if re.search(r"\sv_mov_b32", l):
        b[k][3] += 1
print("line  [pk_add, scratch_store, scratch_load, v_mov]")
for k in sorted(b):
    if b[k][0] or b[k][1] or b[k][2]:
        print(k * 250, b[k])
for i, l in enumerate(body):
    if "scratch_" in l and i < 6000:
        print(i, l.strip()[:90])
