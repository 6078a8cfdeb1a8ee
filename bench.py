#!/usr/bin/env python3
"""bench.py -- GCUPS of the SSV hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]

With --gpus N > 1 and no launcher around it (WORLD_SIZE unset) this script starts its own N workers
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`, as a child process, before
anything in this process has touched a GPU) and exits with their code; under a launcher it is one rank.

A step is one whole pass of the hot path over one batch of synthetic input that
is already resident in HBM: model padding copy + SSV kernel + hit compaction +
ordering of the hit records into the reference's device order (+ for N > 1 the
RCCL gather of the records to rank 0, where the rank lists concatenate to the
ordered whole).  Three passes are in flight (--pipeline-depth), each with its own
context and hit buffer, and each -- from its first kernel to the ordering of its records -- on one of two high-priority
streams that consecutive passes alternate between: while the host waits for pass k and its records are ordered and
gathered, the SSV kernel of pass k+1 runs, and has started while kernel k was draining (a launch's last, half-empty round
of tiles and the gap between two dependent launches are filled by its neighbour: a step then takes LESS than one kernel
alone).  The engine is libhavac_dev.so's pipe (include/havac_dev.h level 2b); the timed region runs inside havac_pipe_run, no
Python between two passes: what is measured is what a C++ caller of the library gets (havac_benchmark --raw --repeat measures
the same through the handle API, without Python in the process).  All K passes are complete when the timed region ends.
`kernel.avg_ms` and `roofline` are the kernel ALONE: HIP events around the launch in the same K passes run strictly one after
the other (`config.ms_per_step_strictly_serial`, the figure of the reference's one-run-at-a-time API), where nothing shares the
chip with it; `kernel.avg_ms_overlapped` is what the same events read inside the timed region.

Workloads (BASELINE.json configs):
  c2 (default)  one pHMM of L = 1024 rows x 100 Mbp (100,012,032 columns after padding to 12288) per GPU; N > 1 is
                WEAK-scaled: the database grows to N x 100,012,032 columns
  c3            the 1000-model collection (lengths log-uniform in [50, 2000], 503,329 rows concatenated as
                host/phmm/PhmmPreprocessor.cpp:9-31 does) x 10 Mbp (10,002,432 columns) per GPU; weak-scaled
  c4            the same collection x 1 Gbp (81,381 segments = 1,000,009,728 columns) in all, STRONG-scaled: the
                database is cut into N runs of whole segments, one per rank
  c5            one pHMM of L = 20000 rows x 100 Mbp per GPU; weak-scaled
The database is cut into N runs of whole 12288-column segments, one per rank (havac_amd/dist.py; a rank recomputes a
left halo of rows-1 columns); a rank keeps in its HBM only the columns its shard reads (havac_ssv_shard_window: its own,
the halo, a few thousand for the tiling -- C4: 31 MB of the 250 MB), and there is no data-path collective.
For N > 1 rank 0 checks the gathered list before it prints (`distributed.parity`): device order, no duplicates, the
ranks' counts add up, every rank's records lie inside its columns, and the records around two shard boundaries equal
the CPU checker's.

GCUPS = defined DP cells (columns x rows; padding outside the matrix and halo recomputation are not
counted) / wall time of the K timed steps, max over ranks.

One JSON line is printed by rank 0; see DESIGN.md section 5 for the extra objects `roofline` (integer-VALU bound,
with the HBM figures next to it; `traffic` = HBM bytes per launch measured by two rocprofv3 --pmc passes of a
torch-free child of this script, started before this process touches the GPU) and `cpu_baseline` (the reference's
softSsv, or our C restatement of it, on the host cores, on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from havac_amd import synth  # noqa: E402

SEGMENT = synth.SEGMENT
FPGA_GCUPS = 1739.0                    # reference README.md:4 (Alveo U50), BASELINE.md section 1

# name -> (rows or None for the 1000-model collection, real symbols per unit, scaling); a unit is one GPU's share
# for weak scaling and the whole database for strong scaling
WORKLOADS = {
    # (the GPU needs ~10 launches of 2 ms to reach its clock under this load -- the kernel's duration falls from 2.11 to 1.83 ms
    # over the first ten launches of a run, profiles/r02b_kernel_stats_c2.csv; bench.py runs its own clock warm-up in front
    # of the --warmup passes and reports how many passes that took: `clock_warmup_passes`)
    "c2": dict(rows=1024, real=100_000_000, scaling="weak", steps=100, warmup=5,
               label="C2: 1 pHMM L=1024 x 100 Mbp (100,012,032 columns padded to 12288) per GPU"),
    "c3": dict(rows=None, real=10_000_000, scaling="weak", steps=10, warmup=2,
               label="C3: 1000-model collection (L 50-2000, 503,329 rows concatenated) x 10 Mbp (10,002,432 columns) per GPU"),
    "c4": dict(rows=None, real=1_000_000_000, scaling="strong", steps=2, warmup=1,
               label="C4: 1000-model collection (503,329 rows) x 1 Gbp (1,000,009,728 columns) sharded over the GPUs"),
    "c5": dict(rows=20000, real=100_000_000, scaling="weak", steps=10, warmup=2,
               label="C5: 1 pHMM L=20000 x 100 Mbp (100,012,032 columns) per GPU"),
}

# Roofline of ssv_diag_kernel.  Not HBM, not MFMA: the kernel is bound by
# VALU issue.  Per DP cell the recurrence needs one 4:1 score select and one saturating add.  The select is served
# by LDS (one conflict-free ds_read_b64 returns the match words of 4 cells), so the only algorithmic VALU work left
# is the add: v_pk_add_i16 with clamp, 2 cells per lane-instruction; gfx950 has no wider saturating add.
# v_pk_add_i16 belongs to the HALF-rate VALU class: a wave64 instruction holds its SIMD for 4 cycles (16 lanes/clk),
# measured with tools/valu_rates.hip (profiles/r01_valu_rates_4waves_per_simd.txt; the full-rate class -- v_add_u32,
# v_and_b32, v_fma_f32 ... -- has no packed saturating add).  Peak for this class:
# 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3e12 lane-instructions/s = 78.6e12 int16 adds/s.
PEAK_TIOPS_I16 = 78.6
PEAK_TIOPS_I16_FULL_RATE_CLASS = 157.3  # the same count against 32 lanes/clk (the fp32-FMA class), for reference
OPS_PER_CELL = 1                        # one int16 saturating add per cell on the VALU
# LDS side of the same kernel: 8 bytes (one ds_read_b64) per lane per 4 cells = 2 B/cell;
# peak 256 B/clk/CU (MI355X_MICROARCH.md, LDS table) x 256 CUs x 2.4 GHz = 157.3 TB/s.
LDS_BYTES_PER_CELL = 2
LDS_PEAK_TBS = 157.3
HBM_PEAK_GBS = 8000.0


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def plant_packed(packed: np.ndarray, consensus: np.ndarray, nreal: int, every=1_000_000, length=300, sub=0.15,
                 seed=synth.SEED_PLANT) -> int:
    """synth.plant_homologs on a 2-bit packed buffer (only the touched bytes are unpacked)."""
    rng = np.random.default_rng(seed)
    length = min(length, consensus.size)
    planted, pos = 0, every // 2
    while pos + length <= nreal:
        start = int(rng.integers(0, consensus.size - length + 1))
        piece = consensus[start:start + length].copy()
        mut = rng.random(length) < sub
        piece[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        b0, b1 = pos // 4, (pos + length + 3) // 4
        sym = synth.unpack_2bit(packed[b0:b1])
        sym[pos - 4 * b0: pos - 4 * b0 + length] = piece
        packed[b0:b1] = synth.pack_2bit(sym)
        planted += 1
        pos += every
    return planted


def make_inputs(workload: str, world: int, rows_override=None, columns_override=None):
    """-> (model int8 [rows,4], packed sequence, total columns, columns per GPU, planted homologs); same on every rank"""
    w = WORKLOADS[workload]
    if rows_override:
        model, consensus = synth.dfam_like_model(rows_override, synth.SEED_MODEL)
    elif w["rows"] is None:
        model, consensus = synth.model_collection(synth.model_lengths(1000), synth.SEED_MODEL)
    else:
        model, consensus = synth.dfam_like_model(w["rows"], synth.SEED_MODEL)
    unit = synth.padded_length(w["real"]) if not columns_override else columns_override
    assert unit % SEGMENT == 0
    ncols = unit if w["scaling"] == "strong" else unit * world
    # the real symbols of every unit: what the padding to whole segments adds is symbol 0 ('A',
    # host/sequence/SequencePreprocessor.cpp:41) at the very end of the database
    nreal = ncols - (unit - w["real"]) * (1 if w["scaling"] == "strong" else world) if not columns_override else ncols
    packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
    planted = plant_packed(packed, consensus, nreal)
    packed[nreal // 4:] = 0
    return model, packed, ncols, (ncols // world if w["scaling"] == "strong" else unit), planted


# ---- HBM traffic of the SSV kernel from the hardware counters -----------------------------------------------------
def pmc_child(args):
    """The program rocprofv3 --pmc runs: two passes of the workload through the handle API (ctypes + numpy only, no
    torch: the profiler's library initialises the GPU before this program starts, so it must not re-exec anything)."""
    from havac_amd.hw_client import HavacHwClient
    model, packed, ncols, _, _ = make_inputs(args.workload, 1, args.rows, args.columns_per_gpu)
    client = HavacHwClient(deviceIndex=0)
    client.setHitCapacity(max(1 << 20, int(ncols * model.shape[0] * 1.2e-5)))
    client.writeSequence(packed)
    client.writePhmm(model)
    for _ in range(2):
        client.invokeHavacSsvAsync()
        client.waitForHavacSsvAsync()
    print("pmc child done", client.getNumHits(), flush=True)


def measure_traffic(args):
    """HBM bytes per launch of ssv_diag_kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (they do
    not fit one pass on gfx950: MI355X_MICROARCH.md, PMC slots), traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 with the
    guide's gfx950 FETCH_SIZE correction.  Must run before this process initialises the GPU.  -> (bytes or None, note)"""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    out = tempfile.mkdtemp(prefix="havac_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    values = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", counter.lower(), "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload]
            if args.rows:
                cmd += ["--rows", str(args.rows)]
            if args.columns_per_gpu:
                cmd += ["--columns-per-gpu", str(args.columns_per_gpu)]
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=args.pmc_timeout)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}: {(r.stderr or r.stdout)[-300:]}"
            per_dispatch = {}
            for path in glob.glob(os.path.join(out, "**", f"*{counter.lower()}*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if ("ssv_diag_kernel" in row["Kernel_Name"] or "ssv_resident_kernel" in row["Kernel_Name"]) and row["Counter_Name"] == counter:
                            per_dispatch[row["Dispatch_Id"]] = per_dispatch.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per_dispatch:
                return None, f"no ssv_diag_kernel / ssv_resident_kernel rows in the {counter} pass"
            values[counter] = sum(per_dispatch.values()) / len(per_dispatch)
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 --pmc pass timed out"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    traffic = (2.0 * values["FETCH_SIZE"] + values["WRITE_SIZE"]) * 1024.0
    return traffic, (f"rocprofv3 --pmc, two passes of a child of this script: FETCH_SIZE {values['FETCH_SIZE']:.1f} KB "
                     f"(x2: gfx950 counts half the bytes of 4-, 8- and 16-byte-per-lane loads alike, tools/fetch_calibration.hip), WRITE_SIZE {values['WRITE_SIZE']:.1f} KB "
                     "per launch")


# ---- CPU baseline -------------------------------------------------------------------------------------------------
def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            names = [line.split(":", 1)[1].strip() for line in f if line.startswith("model name")]
        return f"{names[0]} ({len(names)} logical CPUs visible)" if names else "unknown"
    except OSError:
        return "unknown"


def cpu_baseline(packed: np.ndarray, model: np.ndarray, gpu_hits, cores: int, whole_list_limit_cells=6e12):
    """Time the CPU path on a bounded sample -- the first columns of the database against the first `R` rows of the
    model (all of them up to 20000 rows; a 4096-row prefix of a taller collection, which is self-contained because
    a diagonal's score at row r depends on rows <= r only) -- and check that it finds exactly the records the GPU
    reported for those cells.  gpu_hits: callable (col_lo, col_hi, row_hi) -> the GPU's records in that range."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as O
    nrows = model.shape[0]
    R = nrows if nrows <= 20000 else 4096
    cols_per_core = int(min(500_000, max(100_000, 6e9 / R))) // 4 * 4
    sample_cols = min(cores * cols_per_core, packed.size * 4)
    sample_cols -= sample_cols % 4
    sym = synth.unpack_2bit(packed[: sample_cols // 4])
    sub = np.ascontiguousarray(model[:R])
    use_ref = O.ref_available()
    blocks = [(k * cols_per_core, min((k + 1) * cols_per_core, sample_cols)) for k in range(cores)]
    blocks = [b for b in blocks if b[0] < b[1]]

    def one(block):
        a, b = block
        start = max(0, a - (R - 1))
        if use_ref:     # the reference's own softSsvThreshold256 on [start, b), hits left of `a` dropped
            h = O.ssv_reference(sym[start:b], sub)
            rows, cols = O.unpack_hits(h)
            cols = cols + np.uint64(start)
            keep = cols >= np.uint64(a)
            return O.pack_hits(rows[keep], cols[keep])
        return O.ssv_window(sym, sub, a, b)

    O.lib()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(blocks)) as pool:
        parts = list(pool.map(one, blocks))
    dt = time.perf_counter() - t0
    cpu_hits = O.device_order(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)
    match = bool(np.array_equal(cpu_hits, O.device_order(gpu_hits(0, sample_cols, R))))
    cells = sample_cols * R
    out = {
        "value": round(cells / dt / 1e9, 4), "unit": "GCUPS", "cores": len(blocks),
        "kind": "reference" if use_ref else "port",
        "sample": f"first {sample_cols} columns x first {R} of {nrows} rows of the same workload ({cells:.3g} cells, {dt:.1f} s wall, "
                  f"{len(blocks)} threads each a column block with a {R - 1}-column left halo)",
        "hits_match_gpu": match, "hits_in_sample": int(cpu_hits.size),
        # the CPU and the build of what was timed, next to the core count
        "cpu_model": cpu_model(), "compiler": O.build_info(reference=use_ref),
        "compiler_vectorised_port": O.build_info(reference=False),
    }
    # SURVEY.md 8d's other CPU figure: the reference-shaped loop nest (test/softSsv/SoftSsv.cpp:31-62) on ONE core
    one_cols = int(min(sample_cols, max(40_000, 1.5e9 / R))) // 4 * 4
    t1 = time.perf_counter()
    one_hits = O.ssv_reference(sym[:one_cols], sub) if use_ref else O.ssv(sym[:one_cols], sub)
    dt_one = time.perf_counter() - t1
    out["single_thread"] = {
        "value": round(one_cols * R / dt_one / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": "reference" if use_ref else "port",
        "sample": f"first {one_cols} columns x first {R} rows ({one_cols * R:.3g} cells, {dt_one:.1f} s wall), one thread, the "
                  "reference's loop nest (rows outer, columns inner, one u8 row buffer)",
        "hits_match_gpu": bool(np.array_equal(O.device_order(one_hits), O.device_order(gpu_hits(0, one_cols, R))))}
    ncols = packed.size * 4
    if ncols * nrows <= whole_list_limit_cells:
        # second CPU figure: our AVX2 restatement (oracle.ssv_fast) over the WHOLE workload, which also lets the bench
        # compare the complete hit list of the timed launch, not only the sample's
        whole = synth.unpack_2bit(packed)
        mine = gpu_hits(0, ncols, nrows)
        t1 = time.perf_counter()
        fast_hits = O.ssv_fast(whole, model, nthreads=cores, cap=max(1 << 20, mine.size + 1024))
        dt_fast = time.perf_counter() - t1
        out["vectorised_port"] = {
            "value": round(whole.size * nrows / dt_fast / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": f"the whole workload, {whole.size} columns x {nrows} rows, AVX2 int16 lanes, {dt_fast:.1f} s wall",
            "whole_hit_list_matches_gpu": bool(np.array_equal(fast_hits, mine))}
    else:
        # too many cells for any CPU route: one stretch of 5e6 columns over the whole height, every record
        a = (ncols // 2) // SEGMENT * SEGMENT
        b = a + 5_000_000
        start = max(0, a - (nrows - 1)) // 4 * 4
        stretch = synth.unpack_2bit(packed[start // 4: b // 4])
        mine = gpu_hits(a, b, nrows)
        t1 = time.perf_counter()
        want = O.ssv_fast(stretch, model, nthreads=cores, cap=mine.size + (1 << 24))
        rows_w, cols_w = O.unpack_hits(want)
        keep = cols_w + np.uint64(start) >= np.uint64(a)
        want = O.device_order(O.pack_hits(rows_w[keep], cols_w[keep] + np.uint64(start)))
        dt_fast = time.perf_counter() - t1
        out["vectorised_port"] = {
            "value": round(stretch.size * nrows / dt_fast / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": f"columns [{a}, {b}) x all {nrows} rows with their {nrows - 1}-column halo, AVX2 int16 lanes, {dt_fast:.1f} s wall",
            "records_in_stretch": int(want.size), "stretch_matches_gpu": bool(np.array_equal(want, mine))}
    return out


def describe_plan(ncols, nrows, rank, world, tuning, wave_slots=0):
    """how this rank's launch hands out its tiles (havac_ssv_plan: the planner havac_ssv_enqueue uses, with the wave slots
    of the context that launched: havac_ssv_wave_slots), for the JSON line"""
    from havac_amd.ssv import launch_plan
    t = list(tuning or [])
    t = [t[0] if len(t) > 0 else -1, t[1] if len(t) > 1 else -1, -1, -1] + (t[4:9] if len(t) > 4 else [])
    p = launch_plan(ncols, nrows, rank, world, wave_slots=wave_slots, tuning=t)
    cut = p["cut_tiles"] if p["nrow_blocks"] > 1 else 0
    if p["resident_kernel"]:
        return {"kernel": "ssv_resident_kernel (tables resident in LDS, every wave walks its run of tiles)", "wave_slots": wave_slots or 256 * 24,
                "tiles": p["ntiles"], "workgroups": p["workgroups"], "waves_per_round": p["walk_slots"], "tiles_per_wave_by_round": p["walk_len"]}
    return {"kernel": "ssv_diag_kernel",
            "tiles_per_walk": p["tiles_per_group"], "wave_slots": wave_slots or 256 * 24, "tiles": p["ntiles"], "partitions": p["nparts"], "workgroups": p["workgroups"],
            "cut_tiles_per_partition": "all" if cut >= max(b - a for a, b in zip(p["part_begin"], p["part_begin"][1:])) and cut else cut,
            "row_blocks_per_cut_tile": p["nrow_blocks"] if cut else 0,
            "row_blocks": [list(b) for b in p["row_blocks"][:4]] + (["..."] if len(p["row_blocks"]) > 4 else [])}


class OrderedHits:
    """The ordered record list of a pass (a torch int64 tensor on the device, device order = segment-major) with the
    lookups the checks need."""

    def __init__(self, merged, nrows):
        self.merged, self.nrows, self.n = merged, nrows, int(merged.numel())

    def first_of_segment(self, seg):
        """index of the first record of segment >= seg"""
        lo_i, hi_i = 0, self.n
        while lo_i < hi_i:
            mid = (lo_i + hi_i) // 2
            if ((int(self.merged[mid].item()) >> 14) & 0x3FFFFFF) < seg:
                lo_i = mid + 1
            else:
                hi_i = mid
        return lo_i

    def window(self, col_lo, col_hi, row_hi):
        """the records with col_lo <= column < col_hi and row < row_hi: the segments that hold those columns are one
        contiguous slice of the ordered list; the rest is filtered on the device"""
        part = self.merged[self.first_of_segment(col_lo // SEGMENT): self.first_of_segment((col_hi + SEGMENT - 1) // SEGMENT)]
        cols = ((part >> 14) & 0x3FFFFFF) * SEGMENT + (part & 0x3FFF)
        keep = (cols >= col_lo) & (cols < col_hi)
        if row_hi < self.nrows:
            keep &= ((part >> 40) & 0xFFFFFF) < row_hi
        return part[keep].cpu().numpy().view(np.uint64)


def gathered_list_checks(merged, counts, spans):
    """Order, duplicates, counts and columns-per-rank of a gathered list, on the device and through the C ABI
    (havac_ssv_check_order: one grid-stride HIP pass, no list-sized temporary, 64-bit indices).  Not torch: C4's list is
    4.46e9 records -- beyond the 2^32 elements up to which this torch build's kernels index correctly
    (tools/big_sort_check.py) -- and 36 GB per full-size temporary on a card that already holds the receive buffers."""
    import torch
    from havac_amd.ssv import check_order
    n = int(merged.numel())
    out = {"records": n, "counts_add_up": int(sum(counts)) == n}
    if not out["counts_add_up"]:
        out["device_order_no_duplicates"] = out["ranks_inside_their_columns"] = None
        return out
    torch.cuda.current_stream(merged.device).synchronize()
    rep = check_order(merged.data_ptr(), n, counts, spans)
    out["device_order_no_duplicates"] = rep["out_of_order"] == 0
    out["ranks_inside_their_columns"] = rep["out_of_span"] == 0
    if rep["out_of_order"]:
        out["out_of_order"] = {"records": rep["out_of_order"], "first_index": rep["first_out_of_order"]}
    if rep["out_of_span"]:
        out["outside_their_rank"] = {"records": rep["out_of_span"], "first_index": rep["first_out_of_span"]}
    return out


def distributed_parity(hits: OrderedHits, counts, spans, packed, model, cores, stretch=200_000):
    """What rank 0 checks on the gathered list of an N > 1 run before it prints: the list is in the reference's device
    order without duplicates; the ranks' counts add up to it; every rank's records lie inside that rank's columns
    (gathered_list_checks: one HIP pass over the list); and around (up to) two shard boundaries -- `stretch` columns on
    either side: the cut side of the left rank, the halo side of the right one -- the records equal the CPU checker's,
    every one of them (slices of a few 10^5 records, looked up by bisection)."""
    from oracle import pyoracle as O
    out = gathered_list_checks(hits.merged, counts, spans)
    nrows, ncols = model.shape[0], packed.size * 4
    boundaries = sorted({spans[0][1], spans[-1][0]}) if len(spans) > 1 else []
    checked = []
    for cut in boundaries:
        a, b = max(0, cut - stretch), min(ncols, cut + stretch)
        start = max(0, a - (nrows - 1)) // 4 * 4
        sym = synth.unpack_2bit(packed[start // 4: (b + 3) // 4])[: b - start]
        mine = hits.window(a, b, nrows)
        want = O.ssv_fast(sym, model, nthreads=cores, cap=mine.size + (1 << 22))
        rows_w, cols_w = O.unpack_hits(want)
        keep = cols_w + np.uint64(start) >= np.uint64(a)
        want = O.device_order(O.pack_hits(rows_w[keep], cols_w[keep] + np.uint64(start)))
        checked.append({"boundary_column": int(cut), "columns": [int(a), int(b)], "records": int(want.size),
                        "equals_cpu_checker": bool(np.array_equal(want, mine))})
    out["boundary_stretches"] = checked
    out["ok"] = bool(out["counts_add_up"] and out["device_order_no_duplicates"] and out["ranks_inside_their_columns"] and
                     all(c["equals_cpu_checker"] for c in checked))
    return out


class Watchdog:
    """N > 1: a rank that stays in one stage longer than the deadline says which rank and which stage on stderr and ends the
    process with a non-zero code (os._exit: a fresh exit, nothing re-executed) -- under torch.distributed.run that ends the
    job.  Without it a rank that died, or a collective a peer never reaches, leaves the other ranks waiting until the
    driver's own limit.  The C-ABI gather has the same deadline on its own host waits (havac_gather_set_deadline)."""

    def __init__(self, rank: int, limit_s: float):
        import threading
        self.rank, self.limit, self.name, self.since = rank, limit_s, "start", time.monotonic()
        self.stage_limit = limit_s
        self._stop = threading.Event()
        if limit_s > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def stage(self, name: str, limit_s: float | None = None):
        self.name, self.since, self.stage_limit = name, time.monotonic(), (self.limit if limit_s is None else limit_s)

    def stop(self):
        self._stop.set()

    def _watch(self):
        while not self._stop.wait(0.5):
            waited = time.monotonic() - self.since
            if self.limit > 0 and waited > self.stage_limit:
                print(f"bench.py: rank {self.rank} has been in stage '{self.name}' for {waited:.0f} s (deadline {self.stage_limit:.0f} s): "
                      "giving up, exit code 3", file=sys.stderr, flush=True)
                os._exit(3)


def launch_workers(args, argv):
    """python bench.py --gpus N without a launcher: start N ranks as a child process (nothing in this process has
    touched a GPU), pass their output through, exit with their code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--rows", type=int, default=0, help="probe: one Dfam-like model of this many rows instead of the workload's")
    ap.add_argument("--columns-per-gpu", type=int, default=0, help="probe: this many columns (a multiple of 12288) per unit")
    ap.add_argument("--pipeline-depth", type=int, default=0,
                    help="passes in flight (own context, hit buffer and stream each): the ordering / gather of pass k "
                         "overlaps the SSV kernel of pass k+1, which starts on the other of two streams "
                         "while kernel k drains.  1 = strictly serial; 0 = 3 (the third is slack for a host that wakes late from "
                         "its wait; 2 for C4 on several GPUs), 1 above 1e14 cells on one GPU")
    ap.add_argument("--tuning", default="", help="experiments: rows_per_block,tiles_per_item,block_tails,ordering[,parts_log2,"
                    "split_rounds_x4,short_rows,guide] for havac_ssv_set_tuning / havac_ssv_set_split_tuning (-1 = the library's own "
                    "rule), e.g. --tuning=-1,-1,-1,0 orders with the radix sort")
    ap.add_argument("--kernel-streams", type=int, default=0, choices=(0, 1, 2, 3, 4),
                    help="streams consecutive passes take in turn (a pass -- preparation, SSV kernel, ordering -- is one stream's "
                         "business): 0 = the library's rule (two where passes are in flight), 1 = every pass on one stream, "
                         "2 .. 4 = that many (three and four: experiments)")
    ap.add_argument("--backend", default=os.environ.get("HAVAC_BENCH_BACKEND", "nccl"), choices=("nccl", "gloo"),
                    help="torch.distributed backend of an N > 1 run: nccl (= RCCL, one rank per GPU: what the driver runs) or gloo (a "
                         "rehearsal with several ranks on ONE GPU, which RCCL refuses)")
    ap.add_argument("--gather", default="c_abi", choices=("c_abi", "torch"),
                    help="N > 1: the records travel through libhavac_dev.so's own RCCL calls (havac_gather_*) or through torch.distributed")
    ap.add_argument("--gather-library", default="",
                    help="rehearsals: a stand-in for librccl.so.1 (tests/native/rccl_standin.cpp) handed to havac_gather_use_library, so that "
                         "the C-ABI gather runs between ranks that share a GPU; with --backend gloo")
    ap.add_argument("--force-dist", action="store_true", default=os.environ.get("HAVAC_BENCH_FORCE_DIST") == "1",
                    help="run the N > 1 code path (process group, gather, barrier) with the ranks there are, even one")
    ap.add_argument("--deadline", type=float, default=180.0,
                    help="N > 1: seconds a rank may spend in one stage (a pass with its gather, a barrier, a check) before it says which "
                         "rank and stage and exits non-zero; also the deadline of every host wait of the C-ABI gather.  0 = none")
    ap.add_argument("--die-at", default="", help=argparse.SUPPRESS)     # tests: "RANK:PASS[:hang]" -- before that pass of the timed region that rank leaves (os._exit), or stops making calls
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="N > 1: skip rank 0's check of the gathered list (distributed.parity)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--pmc-timeout", type=int, default=240)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    w = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = w["steps"]
    if args.warmup is None:
        args.warmup = w["warmup"]
    if args.pmc_child:
        return pmc_child(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # --force-dist: rehearse the N > 1 code path (process group, gather, barrier, all-reduce) with the
    # ranks there are, even one -- the only way to run it over RCCL on a 1-GPU box
    use_dist = world > 1 or args.force_dist
    die_at = tuple(int(v) if v.isdigit() else v for v in args.die_at.split(":")) if args.die_at else None
    watchdog = Watchdog(rank, args.deadline if use_dist else 0.0)

    # HBM traffic from the counters, by a child under rocprofv3, BEFORE this process initialises the GPU
    traffic, traffic_note = None, "not measured (N > 1, --no-pmc, or a pass of more than 1e13 cells)"
    model, packed, ncols, cols_per_gpu, planted = make_inputs(args.workload, world, args.rows, args.columns_per_gpu)
    nrows = model.shape[0]
    if world == 1 and not use_dist and not args.no_pmc and ncols * nrows <= 1e13:
        traffic, traffic_note = measure_traffic(args)

    import torch
    import torch.distributed as dist
    from havac_amd.dist import ShardedSsv
    from havac_amd.ssv import shard_cells, shard_columns

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # --backend gloo lets several ranks share one GPU for a rehearsal of the N>1 path on a 1-GPU box
    # (RCCL refuses two ranks on one device); the driver's real multi-GPU runs use the default, nccl = RCCL.
    backend = args.backend
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500))
        watchdog.stage("process group")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        from havac_amd import dist as hdist
        hdist.set_gather_route(args.gather)
        if args.gather_library:
            hdist.use_gather_library(os.path.abspath(args.gather_library))
        hdist.set_gather_deadline(args.deadline)

    from havac_amd.ssv import shard_window
    # a rank holds only the columns its shard reads (N > 1); the host copy stays whole for the checks behind the timed region
    win_first, win_end = shard_window(ncols, nrows, rank, world) if world > 1 else (0, ncols)
    d_seq = torch.from_numpy(packed[win_first // 4: win_end // 4]).to(device)
    d_phmm = torch.from_numpy(model.reshape(-1)).to(device)

    my_cells = shard_cells(ncols, nrows, rank, world)
    total_cells = ncols * nrows
    # records per cell: 1.0e-5 on C2, 0.9e-5 on the collection
    hit_capacity = max(1 << 20, int(my_cells * (4e-5 if my_cells <= 1e13 else 1.2e-5)))
    # kernels of consecutive passes side by side (two kernel streams) where a pass is long enough to gain from it (512 rows x
    # 100 Mbp, 5.1e10 cells: +1.8 %; 1024 rows: +3.5 %; 256 rows, 2.6e10 cells, the resident-table kernel: -4 %)
    # (the rule itself lives in libhavac_dev.so's pipe; --kernel-streams 1 / 2 force one or the other)
    kernel_streams = args.kernel_streams or None
    # passes in flight: 3 -- the next pass's kernel starts (on the other of two streams) while this one drains, and the host's
    # wait, the ordering and for N > 1 the gather (C4: 36 GB to rank 0 per pass) hide behind it.  Two are enough for that on a quiet
    # host (C2 57.25-57.27 TCUPS with two, 57.27-57.31 with three; short passes gain from the third: 64 rows 43.7 -> 44.4, 32 rows
    # 32.7 -> 37.1: profiles/r05n_*); the third is slack for a host that wakes late from its wait for pass k -- on a box whose
    # CPUs were busy a two-deep run lost 10 % to that (profiles/r05o_bench_c2_busy_host.json).  1 above 1e14 cells on one GPU (C4
    # on one card: 93 GB of hit and ordering buffers per pass in flight)
    if args.pipeline_depth > 0:
        depth = args.pipeline_depth
    else:
        depth = 3 if world > 1 or my_cells <= 1e14 else 1
        if world > 1 and total_cells > 1e14:      # (C4: rank 0 holds a 36 GB receive buffer per pass in flight)
            depth = 2
    tuning = [int(v) for v in args.tuning.split(",")] if args.tuning else None
    engine = ShardedSsv(hit_capacity, device, depth=depth, gather_when_alone=use_dist, tuning=tuning, kernel_streams=kernel_streams)
    if world > 1:
        engine.set_sequence_window(win_first, win_end - win_first)
    wave_slots = engine.ctx.wave_slots()

    def fence(stage="fence"):
        # What this rank has queued is complete BEFORE torch's collective starts: the gather's communicator (libhavac_dev.so's own)
        # and torch's must never have operations in flight side by side (two communicators used concurrently can hang); the
        # records still travelling behind the last pass are waited for with the gather's deadline, not inside a synchronize
        # that a dead peer would never let return.
        watchdog.stage(stage)
        engine_now.wait_gathers()
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    def run_steps(engine, nsteps, stage="passes"):
        """`nsteps` whole passes, all finished on return -> (last result, per-pass (kernel ms, enqueue-to-ordered ms)).  The
        loop -- submit, and collect the oldest once every slot is in flight -- runs inside libhavac_dev.so (havac_pipe_run): no
        Python between two passes, the timed region is what a C++ caller of the library gets."""
        if die_at and stage == "timed region" and die_at[0] == rank:      # (tests only)
            before = die_at[1] - 1
            result, timings = engine.run_many(before, d_seq, ncols, d_phmm, nrows, inputs_ready=True) if before > 0 else (None, [])
            watchdog.stage(f"{stage}: pass {die_at[1]} of {nsteps}")
            if len(die_at) > 2:
                time.sleep(3600)         # a rank that hangs: its own watchdog, and every peer's, must end the job
            os._exit(17)                 # a rank that dies without a word
        # (the deadline of a stage that holds `nsteps` passes: the per-stage one, and room for the passes themselves)
        watchdog.stage(f"{stage}: {nsteps} pass(es)", limit_s=(args.deadline + nsteps * pass_seconds * 3) if args.deadline else None)
        return engine.run_many(nsteps, d_seq, ncols, d_phmm, nrows, inputs_ready=True)

    # set-up, not warm-up: one pass through every slot so that each context has its sort buffers before anything is
    # timed (a slot first used inside the timed region would pay a hipMalloc there when --warmup < --pipeline-depth)
    engine_now = engine
    pass_seconds = 1.0
    torch.cuda.synchronize(device)          # the inputs are in place: no pass waits for torch's stream from here on
    t_setup = time.perf_counter()
    run_steps(engine, depth, "set-up")
    pass_seconds = max(1e-4, (time.perf_counter() - t_setup) / depth)
    # Clock warm-up, independent of --warmup: under this load the GPU reaches its clock only after ~10 launches of 2 ms (the
    # kernel's duration falls from 2.11 to 1.83 ms over the first ten launches of a run).  Untimed passes until the kernel's
    # event time has stopped falling -- three passes in a row within 0.5 % of the best seen -- at most 30 passes or 3 s.
    clock_passes, best_ms, steady, t_clock = 0, None, 0, time.perf_counter()
    while True:
        _, t = run_steps(engine, 1, "clock warm-up")
        clock_passes += 1
        k_ms = t[-1][0]
        steady = steady + 1 if (best_ms is not None and k_ms <= best_ms * 1.005) else 0
        best_ms = k_ms if best_ms is None else min(best_ms, k_ms)
        done = steady >= 3 or clock_passes >= 30 or time.perf_counter() - t_clock > 3.0
        if use_dist:      # a pass holds a collective: every rank goes on until all are done
            engine.wait_gathers()
            torch.cuda.synchronize(device)      # (see fence: nothing of the gather's communicator in flight beside torch's)
            flag = torch.tensor([1 if done else 0], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item())
        if done:
            break
    run_steps(engine, args.warmup, "warm-up")
    engine.gather_times()
    engine.gather_ms = []
    fence("barrier in front of the timed region")
    t0 = time.perf_counter()
    (merged, found), kernel_ms = run_steps(engine, args.steps, "timed region")
    fence("barrier behind the timed region")
    elapsed = time.perf_counter() - t0
    gather_ms = list(engine.gather_times())
    serial_ms, serial_timings = None, kernel_ms
    # Everything is on the device and finished (fence): the records of the last pass stay where the gather (N > 1) or the
    # ordering (N = 1) left them -- no copy; `merged` keeps that one buffer alive -- and every other buffer of the pipelined
    # engine is given back before anything else is allocated.  Rank 0 of `--gpus 8 --workload c4` holds 36 GB of records per receive buffer (DESIGN.md section 6
    # has the sum); the checks below allocate nothing of that size.
    engine_variant = engine.ctx.last_kernel_variant()
    kernel_streams = engine.streams_used      # (what the engine really used)
    engine.release()
    if depth > 1:       # the same steps strictly one after the other, for the record (not `value`)
        serial = ShardedSsv(hit_capacity, device, depth=1, gather_when_alone=use_dist, tuning=tuning)
        engine_now = serial
        if world > 1:
            serial.set_sequence_window(win_first, win_end - win_first)
        run_steps(serial, 2, "strictly serial passes, set-up")
        fence("barrier in front of the strictly serial passes")
        t1 = time.perf_counter()
        _, serial_timings = run_steps(serial, args.steps, "strictly serial passes")
        fence("barrier behind the strictly serial passes")
        serial_ms = (time.perf_counter() - t1) / args.steps * 1e3
        serial.release()
        del serial
    torch.cuda.empty_cache()
    per_rank, failed = None, None
    watchdog.stage("per-rank report (all-reduce, all-gather of objects)")
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        lo, hi = shard_columns(ncols, rank, world)
        h = min(nrows - 1, lo)
        halo = h * (h + 1) // 2 + (nrows - 1 - h) * lo     # cells left of its columns a rank recomputes: sum over rows p of min(p, lo)
        mine = {"rank": rank, "device": torch.cuda.get_device_name(device), "columns": [lo, hi], "records": int(found),
                "kernel_ms": round(float(np.mean([k[0] for k in serial_timings])), 4), "halo_cells": int(halo),
                "gather_ms": round(float(np.mean(gather_ms)), 4) if gather_ms else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gcups = total_cells / (elapsed / args.steps) / 1e9
        # the kernel alone: the strictly serial passes' event times (with passes in flight a kernel shares the chip with the
        # head of its successor and the tail of its predecessor, and its events show that)
        ssv_ms = float(np.mean([k[0] for k in serial_timings]))
        ssv_ms_overlapped = float(np.mean([k[0] for k in kernel_ms]))
        enq_ms = float(np.mean([k[1] for k in serial_timings]))   # of the strictly serial passes: no queueing in it
        nhits = int(merged.numel())
        # algorithmic HBM bytes of this rank's launch: its share of the packed sequence once, the
        # model once (4 B/row), 8 B per hit
        algo_bytes = ncols / 4 / world + 4 * nrows + 8 * found
        kernel_s = ssv_ms / 1e3
        achieved_tiops = my_cells * OPS_PER_CELL / kernel_s / 1e12
        custom = bool(args.rows or args.columns_per_gpu)
        out = {
            "metric": "GCUPS (billion SSV cells/s); hit-list bit-exact vs softSsv",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "clock_warmup_passes": clock_passes,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": w["scaling"],
            "vs_baseline": round(gcups / FPGA_GCUPS, 3), "dtype": "i16",
            "data": "synthetic",
            "config": {
                "workload": (("custom probe on " if custom else "") + w["label"] +
                             (f" [--rows {args.rows}]" if args.rows else "") +
                             (f" [--columns-per-gpu {args.columns_per_gpu}]" if args.columns_per_gpu else "") +
                             "; int8 SSV, one kernel launch per step and GPU"),
                "rows": nrows, "columns": ncols, "columns_per_gpu": cols_per_gpu, "cells_per_step": total_cells,
                "hits_per_step": nhits, "planted_homologs": planted, "passes_in_flight": depth, "kernel_streams": kernel_streams,
                "overlap": (None if kernel_streams < 2 else
                            f"consecutive passes take {'two' if kernel_streams == 2 else kernel_streams} streams in turn (a pass -- preparation, SSV kernel, ordering -- on one of them): kernel k+1 starts while kernel k drains and runs beside the ordering of pass k, so ms_per_step can be "
                            "BELOW kernel.avg_ms (the kernel alone, from the strictly serial passes); every one of the K passes is complete "
                            "inside the timed region, and each pass's hit list is checked as before"),
                "ms_per_step_strictly_serial": None if serial_ms is None else round(serial_ms, 4),
                "value_strictly_serial": None if serial_ms is None else round(total_cells / serial_ms / 1e6, 2),
                # which figure a caller of the reference's one-run-at-a-time API sees (host/HavacHwClient.cpp:141-157): run, wait, list
                "api_path": "strictly serial (value_strictly_serial): one run at a time, as the reference's HavacHwClient; `value` keeps "
                            "passes in flight",
                "parallelism": f"column-sharded x{world}" + (f", {'RCCL' if backend == 'nccl' else backend} gather of hit records to rank 0" if use_dist else ""),
                "baseline": "1739 GCUPS = reference README.md:4, 1x Alveo U50 FPGA",
                "work_distribution": describe_plan(ncols, nrows, rank, world, tuning, wave_slots),
            },
            "kernel": {"name": "ssv_resident_kernel" if engine_variant else "ssv_diag_kernel", "avg_ms": round(ssv_ms, 4), "enqueue_to_ordered_ms": round(enq_ms, 4),
                       "gcups_kernel_only": round(my_cells / kernel_s / 1e9, 1),
                       "avg_ms_source": "HIP events on the launch stream, " + ("the K strictly serial passes (kernel alone)" if depth > 1 else "the timed passes (one in flight: kernel alone)"),
                       "avg_ms_overlapped": round(ssv_ms_overlapped, 4) if depth > 1 else None,
                       "kernel_streams": kernel_streams},
            "roofline": {
                "bound": "valu", "achieved": round(achieved_tiops, 2), "peak": PEAK_TIOPS_I16,
                "unit": "Tiop/s (int16 saturating adds, 1 per cell; the score select is served by LDS)",
                "frac": round(achieved_tiops / PEAK_TIOPS_I16, 4),
                # what the chip sustained over the timed region (launches overlapping): this rank's cells x K / wall time
                "sustained": {"achieved": round(my_cells * OPS_PER_CELL / (elapsed / args.steps) / 1e12, 2),
                              "frac": round(my_cells * OPS_PER_CELL / (elapsed / args.steps) / 1e12 / PEAK_TIOPS_I16, 4)},
                "frac_vs_full_rate_class_peak": round(achieved_tiops / PEAK_TIOPS_I16_FULL_RATE_CLASS, 4),
                # SURVEY.md 8d wrote the VALU fraction as GCUPS*1e9*2/3.93e13 (select AND add on the VALU, 32-bit lanes);
                # it exceeds 1 here because the select is served by LDS
                "frac_by_survey_8d_formula": round(my_cells / kernel_s * 2 / 3.93e13, 4),
                "traffic": None if traffic is None else round(traffic, 1), "traffic_source": traffic_note,
                "lds": {"bound": "lds", "achieved": round(my_cells * LDS_BYTES_PER_CELL / kernel_s / 1e12, 2),
                        "peak": LDS_PEAK_TBS, "unit": "TB/s (ds_read_b64, 2 B per cell)",
                        "frac": round(my_cells * LDS_BYTES_PER_CELL / kernel_s / 1e12 / LDS_PEAK_TBS, 4)},
                "hbm": {"bound": "hbm", "achieved": round(algo_bytes / kernel_s / 1e9, 3), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(algo_bytes / kernel_s / 1e9 / HBM_PEAK_GBS, 6),
                        "algorithmic_bytes_per_launch": int(algo_bytes)},
            },
        }
        if use_dist:
            out["distributed"] = {
                "world": dist.get_world_size(), "backend": dist.get_backend(),
                "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                "gather": ("all_gather of the counts + grouped send/recv of exactly count[r] records into one buffer on rank 0" +
                           (", RCCL called by libhavac_dev.so (havac_gather_*)" + (f" [stand-in library {os.path.basename(args.gather_library)}]" if args.gather_library else "")
                            if args.gather == "c_abi" and (backend == "nccl" or args.gather_library) else ", through torch.distributed")),
                "deadline_s": args.deadline,
                "gather_ms_rank0": round(float(np.mean(gather_ms)), 4) if gather_ms else None,
                "per_rank": per_rank,
            }
        cores = min(16, len(os.sched_getaffinity(0)))   # a 1-GPU box's CPU share
        ordered = OrderedHits(merged, nrows)
        watchdog.stage("checks of the list on rank 0 (CPU checker)", limit_s=max(args.deadline, 900.0))
        if use_dist and not args.no_parity_check:
            spans = [tuple(r["columns"]) for r in per_rank]
            out["distributed"]["parity"] = distributed_parity(ordered, [r["records"] for r in per_rank], spans, packed, model, cores)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(packed, model, ordered.window, cores)
        print(json.dumps(out), flush=True)
        # a run whose list failed a check is not a result: the line above says which check, the exit code says so too
        # (a driver that reads only `value` and the exit code must not record it as good)
        parity = out.get("distributed", {}).get("parity")
        base = out.get("cpu_baseline")
        if parity is not None and not parity["ok"]:
            failed = "distributed.parity.ok is false"
        elif base is not None and not (base["hits_match_gpu"] and base["single_thread"]["hits_match_gpu"] and
                                       base["vectorised_port"].get("whole_hit_list_matches_gpu", base["vectorised_port"].get("stretch_matches_gpu"))):
            failed = "cpu_baseline: the CPU checker's records differ from the GPU's"
    if use_dist:
        watchdog.stage("last barrier (rank 0 is checking its list)", limit_s=max(args.deadline, 900.0) + 60.0)
        dist.barrier()
        from havac_amd.dist import close_c_gathers
        close_c_gathers()
        dist.destroy_process_group()
    watchdog.stop()
    if failed:
        sys.exit(f"bench.py: {failed}")


if __name__ == "__main__":
    main()
