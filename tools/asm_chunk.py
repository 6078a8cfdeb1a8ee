"""Instruction counts of the SSV kernel's chunk loop on its USUAL path (no separators, no matrix edge, four-step windows,
no hit), from the ISA that `bash tools/kstat.sh` leaves in build/asm2 (or any `hipcc -S -fno-discard-value-names` output:
the block names are what the regions are found by).   python tools/asm_chunk.py [file.s] [-v] [kernel-symbol-prefix]
tests/test_boundary.py::test_ssv_kernel_resources imports chunk_mix() and holds the build to 512 adds of <= 620.

The usual path is taken to be the path from the chunk loop's header to its back edge with the FEWEST vector
instructions: at a conditional branch to a label a few lines further down (the skip over, or the entry into, a rare
block: a window entry outside the matrix, the middle hit test of an "unsafe" chunk, the symbol fetch at the matrix's
edge) both ways are tried; a conditional branch to a far label leaves for a slow path and is not followed."""
import collections
import os
import re
import sys

sys.setrecursionlimit(100000)
DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "asm2",
                       "havac_dev-hip-amdgcn-amd-amdhsa-gfx950.s")
NEAR = 60          # lines: a rare block is shorter than this
BACK = 40          # lines: the back edge lands on the loop header or on a block this close in front of it (it falls into the header)


def chunk_mix(path=DEFAULT, kernel="_ZN5havac15ssv_diag_kernel", verbose=False):
    """-> {"prologue" | "windows" | "epilogue" | "chunk": Counter(opcode -> count)} of the usual path of `kernel`'s chunk loop"""
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith(kernel)][0]
    stop = [i for i, l in enumerate(lines) if i > start and ".Lfunc_end" in l][0]
    body = lines[start:stop]
    label_at = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    first_window = [i for i, l in enumerate(body) if "expand_for_windowILi0E" in l and l.startswith(".LBB")][0]
    after_windows = [i for i, l in enumerate(body) if "step_windowsI" in l and l.startswith(".LBB")][0]
    header = [i for i, l in enumerate(body) if "Loop Header" in l and i < first_window][-1]

    def branch_target(i):
        t = body[i].split(";")[0].strip()
        if t.startswith("s_branch") or t.startswith("s_cbranch"):
            return label_at.get(t.split()[-1])
        return None

    # the chunk loop's back edge: the last branch behind the windows that goes to the header (or to a block that falls into it);
    # that block is where the usual path starts
    back_edges = [i for i in range(after_windows, len(body)) if (branch_target(i) or 0) and header - BACK <= branch_target(i) <= header]
    loop_end = back_edges[0]
    for i in back_edges:
        if i - loop_end < 400:
            loop_end = i
    loop_top = min(branch_target(i) for i in back_edges if i <= loop_end)

    def instruction(i):
        t = body[i].split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            return None
        return t

    memo = {}

    def flow_exit(t):
        """An if / else as the structurizer leaves it: `s_cbranch .. FLOW` over the `then` side, and in the block FLOW (named
        %Flow..) a second conditional branch over the `else` side -- exactly one of the two sides runs.  -> the line of that second
        branch if line t is such a Flow block, else None."""
        if "%Flow" not in body[t]:
            return None
        for k in range(t + 1, min(t + 4, len(body))):
            ins = instruction(k)
            if ins and ins.startswith("s_cbranch") and label_at.get(ins.split()[-1], 0) > k:
                return k
        return None

    def best(i, skip_at=None, take_at=None):
        """-> (vector instructions, list of instruction indices) of the cheapest way from line i to the back edge; the conditional
        branch at line `take_at` is taken, the one at `skip_at` is not (the two halves of a structurized if / else)"""
        path_here = []
        while True:
            if i in memo and skip_at is None and take_at is None:
                n, rest = memo[i]
                return n + sum(1 for k in path_here if instruction(k).startswith("v_")), path_here + rest
            if i > loop_end:                                             # left the loop: not a way round it
                return 10 ** 9, path_here
            t = instruction(i)
            if t is None:
                i += 1
                continue
            path_here.append(i)
            op = t.split()[0]
            if op == "s_branch" or op.startswith("s_cbranch"):
                target = label_at.get(t.split()[-1])
                if target is not None and loop_top <= target <= header and i > after_windows:
                    break                                                # the back edge
                if target is not None and target < loop_top and op == "s_branch":
                    return 10 ** 9, path_here                            # out of the loop
                if target is not None and op == "s_branch":
                    i = target
                    continue
                if target is not None and i == take_at:
                    i = target
                    take_at = None
                    continue
                if i == skip_at:
                    skip_at = None
                    i += 1
                    continue
                if target is not None and 0 < target - i < NEAR:          # both ways
                    second = flow_exit(target)
                    if second is not None:                              # an if / else: the `then` side and not the `else` side, or the other way round
                        a = best(i + 1, take_at=second)
                        b = best(target, skip_at=second)
                    else:
                        a = best(i + 1)
                        b = best(target)
                    n, rest = a if a[0] <= b[0] else b
                    mine = sum(1 for k in path_here if instruction(k).startswith("v_"))
                    if skip_at is None and take_at is None:
                        memo[path_here[0]] = (n + mine, path_here + rest)
                    return n + mine, path_here + rest
            i += 1
        mine = sum(1 for k in path_here if instruction(k).startswith("v_"))
        return mine, path_here

    _, usual = best(loop_top + 1)
    per_region = collections.defaultdict(collections.Counter)
    for k in usual:
        region = "prologue" if k < first_window else ("windows" if k < after_windows else "epilogue")
        t = instruction(k)
        per_region[region][t.split()[0]] += 1
        if verbose:
            print(f"{region:9s} {t[:100]}")
    everything = collections.Counter()
    for name in ("prologue", "windows", "epilogue"):
        everything.update(per_region[name])
    per_region["chunk"] = everything
    return dict(per_region)


def total(c, prefixes):
    return sum(n for k, n in c.items() if k.startswith(prefixes))


def summary(c):
    return (f"VALU {total(c, ('v_',))}  LDS {total(c, ('ds_',))}  VMEM {total(c, ('global_', 'buffer_', 'scratch_'))}  "
            f"SALU/waits/branches {total(c, ('s_',))}")


if __name__ == "__main__":
    path = next((a for a in sys.argv[1:] if a.endswith(".s")), DEFAULT)
    kernel = next((a for a in sys.argv[1:] if a.startswith("_Z")), "_ZN5havac15ssv_diag_kernel")
    mix = chunk_mix(path, kernel, verbose="-v" in sys.argv)
    for name in ("prologue", "windows", "epilogue"):
        c = mix.get(name, collections.Counter())
        print(f"{name:9s} {summary(c)}")
        print("          ", {k: n for k, n in c.most_common(40) if k.startswith("v_")})
    print(f"{'chunk':9s} {summary(mix['chunk'])}")
