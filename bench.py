#!/usr/bin/env python3
"""bench.py -- GCUPS of the SSV hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one whole pass of the hot path over one batch of synthetic input that
is already resident in HBM: model padding copy + SSV kernel + hit compaction +
ordering of the hit records into the reference's device order (+ for N > 1 the
RCCL gather of the records to rank 0, where the rank lists concatenate to the
ordered whole).  By default three passes are in flight (--pipeline-depth), each with its own
context, hit buffer and HIP stream: while the host waits for the hit count of pass k, orders
its records and gathers them, the SSV kernel of pass k+1 runs; the SSV kernels themselves are
chained back to back, never side by side, so their event-timed durations stay clean.  All K
passes are complete when the timed region ends; `config.ms_per_step_strictly_serial` shows
the same K passes with one in flight.

Workload at N = 1 is BASELINE.json configs[1] ("C2"): one pHMM of L = 1024 rows x
100 Mbp of synthetic sequence (100,012,032 columns after padding to 12288),
int8 scores, one kernel launch.  For N > 1 the run is WEAK-scaled: the database
grows to N x 100,012,032 columns and is cut into N runs of whole 12288-column
segments, one per rank (havac_amd/dist.py; a rank recomputes a left halo of
rows-1 columns); every rank holds the whole packed sequence (N x 25 MB) in its
own HBM, so no data-path collective is needed.

GCUPS = defined DP cells (columns x rows; padding outside the matrix is not
counted) / wall time of the K timed steps, max over ranks.

One JSON line is printed by rank 0; see README/DESIGN.md for the extra objects
`roofline` (integer-VALU bound, with the HBM figures next to it) and
`cpu_baseline` (the reference's softSsv, or our C restatement of it, on the
host cores, on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from havac_amd import synth  # noqa: E402

COLUMNS_PER_GPU = 100_012_032          # 100 Mbp padded to 12288 (8139 segments)
ROWS = 1024
FPGA_GCUPS = 1739.0                    # reference README.md:4 (Alveo U50), BASELINE.md section 1

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
# HBM bytes per launch of ssv_diag_kernel on the default workload, from the separate rocprofv3 --pmc
# passes in profiles/r01f_pmc_c2.csv: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 with the gfx950 FETCH_SIZE
# correction of MI355X_MICROARCH.md (uncalibrated for 8-byte-per-lane loads: an upper bound).
PMC_TRAFFIC_C2_BYTES = (2 * 15039.3 + 8457.0) * 1024


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


def cpu_baseline(packed: np.ndarray, model: np.ndarray, gpu_hits: np.ndarray, cores: int, cols_per_core: int):
    """Time the CPU path on a bounded sample (the first cores*cols_per_core columns, all rows) and
    check that it finds exactly the hits the GPU reported for those columns."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as O
    nrows = model.shape[0]
    sample_cols = min(cores * cols_per_core, packed.size * 4)
    sample_cols -= sample_cols % 4
    sym = synth.unpack_2bit(packed[: sample_cols // 4])
    use_ref = O.ref_available()
    blocks = [(k * cols_per_core, min((k + 1) * cols_per_core, sample_cols)) for k in range(cores)]
    blocks = [b for b in blocks if b[0] < b[1]]

    def one(block):
        a, b = block
        start = max(0, a - (nrows - 1))
        if use_ref:     # the reference's own softSsvThreshold256 on [start, b), hits left of `a` dropped
            h = O.ssv_reference(sym[start:b], model)
            rows, cols = O.unpack_hits(h)
            cols = cols + np.uint64(start)
            keep = cols >= np.uint64(a)
            return O.pack_hits(rows[keep], cols[keep])
        return O.ssv_window(sym, model, a, b)

    O.lib()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(blocks)) as pool:
        parts = list(pool.map(one, blocks))
    dt = time.perf_counter() - t0
    cpu_hits = O.device_order(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)
    _, gcols = O.unpack_hits(gpu_hits)
    match = bool(np.array_equal(cpu_hits, O.device_order(gpu_hits[gcols < np.uint64(sample_cols)])))
    cells = sample_cols * nrows
    # second CPU figure: our AVX2 restatement (oracle.ssv_fast) over the WHOLE workload, which also lets the bench
    # compare the complete hit list of the timed launch, not only the sample's
    whole = synth.unpack_2bit(packed)
    t1 = time.perf_counter()
    fast_hits = O.ssv_fast(whole, model, nthreads=cores, cap=max(1 << 20, 2 * gpu_hits.size))
    dt_fast = time.perf_counter() - t1
    vectorised = {"value": round(whole.size * nrows / dt_fast / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
                  "sample": f"the whole workload, {whole.size} columns x {nrows} rows, AVX2 int16 lanes, {dt_fast:.1f} s wall",
                  "whole_hit_list_matches_gpu": bool(np.array_equal(fast_hits, gpu_hits))}
    return {
        "value": round(cells / dt / 1e9, 4), "unit": "GCUPS", "cores": len(blocks),
        "kind": "reference" if use_ref else "port",
        "sample": f"first {sample_cols} columns x {nrows} rows of the same workload ({cells:.3g} cells, {dt:.1f} s wall, "
                  f"{len(blocks)} threads each a column block with a {nrows - 1}-column left halo)",
        "hits_match_gpu": match, "hits_in_sample": int(cpu_hits.size), "vectorised_port": vectorised,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--columns-per-gpu", type=int, default=COLUMNS_PER_GPU)
    ap.add_argument("--pipeline-depth", type=int, default=0,
                    help="passes in flight (own context, hit buffer and stream each): the ordering / gather of pass k "
                         "overlaps the SSV kernel of pass k+1; the SSV kernels stay back to back.  1 = strictly serial; "
                         "0 = 3 for passes up to 1e12 cells per GPU (C2 is 1e11), else 1: ordering tens of millions of "
                         "records next to the following kernel slows that kernel by more than it hides "
                         "(C3 shape: 303 vs 267 ms per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cols-per-core", type=int, default=500_000)
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the SSV kernel from a separate rocprofv3 --pmc pass (profiles/)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from havac_amd.dist import ShardedSsv
    from havac_amd.ssv import shard_cells

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # HAVAC_BENCH_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the N>1 path on a 1-GPU box
    # (RCCL refuses two ranks on one device); the driver's real multi-GPU runs use the default, nccl = RCCL.
    backend = os.environ.get("HAVAC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # HAVAC_BENCH_FORCE_DIST=1: rehearse the N > 1 code path (process group, gather, barrier, all-reduce) with the
    # ranks there are, even one -- the only way to run it over RCCL on a 1-GPU box
    use_dist = world > 1 or os.environ.get("HAVAC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    assert args.columns_per_gpu % synth.SEGMENT == 0
    ncols = args.columns_per_gpu * world
    nrows = args.rows

    # ---- synthetic inputs (same on every rank: same seeds) -------------------
    model, consensus = synth.dfam_like_model(nrows, synth.SEED_MODEL)
    packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
    nreal = ncols - 12_032 * world if args.columns_per_gpu == COLUMNS_PER_GPU else ncols
    planted = plant_packed(packed, consensus, nreal)
    packed[nreal // 4:] = 0                      # padding is symbol 0 ('A'), SequencePreprocessor.cpp:41
    d_seq = torch.from_numpy(packed).to(device)
    d_phmm = torch.from_numpy(model.reshape(-1)).to(device)

    hit_capacity = max(1 << 20, int(args.columns_per_gpu * nrows * 4e-5))
    depth = args.pipeline_depth if args.pipeline_depth > 0 else (3 if args.columns_per_gpu * nrows <= 1e12 else 1)
    engine = ShardedSsv(hit_capacity, device, depth=depth, gather_when_alone=use_dist)
    my_cells = shard_cells(ncols, nrows, rank, world)
    total_cells = ncols * nrows

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def run_steps(engine, nsteps):
        """`nsteps` whole passes, all finished on return -> (last result, per-pass (kernel ms, enqueue-to-ordered ms))"""
        result, timings = None, []
        for _ in range(nsteps):
            engine.submit(d_seq, ncols, d_phmm, nrows)
            if len(engine.in_flight) == len(engine.slots):
                result = engine.collect()
                timings.append(engine.ctx.last_ms())
        while engine.in_flight:
            result = engine.collect()
            timings.append(engine.ctx.last_ms())
        return result, timings

    # set-up, not warm-up: one pass through every slot so that each context has its sort buffers before anything is
    # timed (a slot first used inside the timed region would pay a hipMalloc there when --warmup < --pipeline-depth)
    run_steps(engine, depth)
    run_steps(engine, args.warmup)
    fence()
    t0 = time.perf_counter()
    (merged, found), kernel_ms = run_steps(engine, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if rank == 0 and merged is not None:
        merged = merged.clone()
    serial_ms, serial_timings = None, kernel_ms
    if depth > 1:       # the same steps strictly one after the other, for the record (not `value`)
        serial = ShardedSsv(hit_capacity, device, depth=1, gather_when_alone=use_dist)
        run_steps(serial, 2)
        fence()
        t1 = time.perf_counter()
        _, serial_timings = run_steps(serial, args.steps)
        fence()
        serial_ms = (time.perf_counter() - t1) / args.steps * 1e3
        serial.close()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gcups = total_cells / (elapsed / args.steps) / 1e9
        ssv_ms = float(np.mean([k[0] for k in kernel_ms]))
        enq_ms = float(np.mean([k[1] for k in serial_timings]))   # of the strictly serial passes: no queueing in it
        nhits = int(merged.numel())
        hits_np = merged.cpu().numpy().view(np.uint64)
        # algorithmic HBM bytes of this rank's launch: its share of the packed sequence once, the
        # model once (4 B/row), 8 B per hit
        algo_bytes = ncols / 4 / world + 4 * nrows + 8 * found
        kernel_s = ssv_ms / 1e3
        achieved_tiops = my_cells * OPS_PER_CELL / kernel_s / 1e12
        out = {
            "metric": "GCUPS (billion SSV cells/s); hit-list bit-exact vs softSsv",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(gcups / FPGA_GCUPS, 3), "dtype": "i16",
            "data": "synthetic",
            "config": {
                "workload": (("C2: " if (nrows == ROWS and args.columns_per_gpu == COLUMNS_PER_GPU) else "custom: ") +
                             f"1 pHMM L={nrows} x {args.columns_per_gpu} columns per GPU "
                             "(C2 = 100 Mbp padded to 12288), int8 SSV, one kernel launch per step"),
                "rows": nrows, "columns": ncols, "cells_per_step": total_cells, "hits_per_step": nhits,
                "planted_homologs": planted, "passes_in_flight": depth,
                "ms_per_step_strictly_serial": None if serial_ms is None else round(serial_ms, 4),
                "parallelism": f"column-sharded x{world}" + (f", {'RCCL' if backend == 'nccl' else backend} gather of hit records to rank 0" if world > 1 else ""),
                "baseline": "1739 GCUPS = reference README.md:4, 1x Alveo U50 FPGA",
            },
            "kernel": {"name": "ssv_diag_kernel", "avg_ms": round(ssv_ms, 4), "enqueue_to_ordered_ms": round(enq_ms, 4),
                       "gcups_kernel_only": round(my_cells / kernel_s / 1e9, 1)},
            "roofline": {
                "bound": "valu", "achieved": round(achieved_tiops, 2), "peak": PEAK_TIOPS_I16,
                "unit": "Tiop/s (int16 saturating adds, 1 per cell; the score select is served by LDS)",
                "frac": round(achieved_tiops / PEAK_TIOPS_I16, 4),
                "frac_vs_full_rate_class_peak": round(achieved_tiops / PEAK_TIOPS_I16_FULL_RATE_CLASS, 4),
                # SURVEY.md 8d wrote the VALU fraction as GCUPS*1e9*2/3.93e13 (select AND add on the VALU, 32-bit lanes);
                # it exceeds 1 here because the select is served by LDS
                "frac_by_survey_8d_formula": round(my_cells / kernel_s * 2 / 3.93e13, 4),
                "traffic": args.traffic_bytes if args.traffic_bytes is not None else (
                    PMC_TRAFFIC_C2_BYTES if (world == 1 and nrows == ROWS and args.columns_per_gpu == COLUMNS_PER_GPU) else None),
                "lds": {"bound": "lds", "achieved": round(my_cells * LDS_BYTES_PER_CELL / kernel_s / 1e12, 2),
                        "peak": LDS_PEAK_TBS, "unit": "TB/s (ds_read_b64, 2 B per cell)",
                        "frac": round(my_cells * LDS_BYTES_PER_CELL / kernel_s / 1e12 / LDS_PEAK_TBS, 4)},
                "hbm": {"bound": "hbm", "achieved": round(algo_bytes / kernel_s / 1e9, 3), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(algo_bytes / kernel_s / 1e9 / HBM_PEAK_GBS, 6),
                        "algorithmic_bytes_per_launch": int(algo_bytes)},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = min(16, len(os.sched_getaffinity(0)))   # a 1-GPU box's CPU share
            out["cpu_baseline"] = cpu_baseline(packed, model, hits_np, cores, args.cpu_cols_per_core)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
