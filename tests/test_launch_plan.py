"""The launch planner (havac_dev.hip: plan_launch, through havac_ssv_plan -- no device needed): whatever the shape and the
tuning, every tile of the launch is covered exactly once -- whole, in a group, or by row blocks that tile its rows -- and the
workgroups of a partition suffice for its items.  The kernel (ssv_kernels.hip.h, "items") decodes the same fields the same way;
the GPU tests (test_gpu_scale.py: test_partitions_..., test_row_cut_table_...) check what it computes."""
import numpy as np
import pytest

from havac_amd import synth
from havac_amd.ssv import launch_plan

SEG = synth.SEGMENT
SHAPES = [  # (columns, rows, shard, shards)
    (100_012_032, 1024, 0, 1),        # C2
    (10_002_432, 503_329, 0, 1),      # C3: fewer tiles than wave slots
    (100_012_032, 20_000, 0, 1),      # C5: whole tiles first, the last 1.5 rounds cut
    (1_000_009_728, 503_329, 5, 8),   # C4, rank 5 of 8
    (100_012_032, 32, 0, 1),          # one-chunk tiles: groups of four, the last round single
    (100_012_032, 100, 0, 1),
    (3 * SEG, 300, 0, 1),             # a handful of tiles: one partition
    (SEG, 1, 0, 1),
    (40 * SEG, 65_500, 1, 3),
    (400 * SEG, 9_000, 0, 1),
]
TUNINGS = [(), (-1, -1, -1, -1, -1, -1, -1, -1, 0), (-1, 2, -1, -1, -1, -1, -1, -1, 1), (-1, -1, -1, -1, 3, 1, 1024, 2), (-1, -1, -1, -1, 0, 64, 2048, 3), (4096, -1, -1, -1, 3), (0,), (-1, 3), (-1, 8, -1, -1, 2), (-1, -4),
           (-1, -1, -1, -1, 3, 6, 1024, 16)]


def check(plan):
    n, rows = plan["ntiles"], plan["nrows_padded"]
    pb = plan["part_begin"]
    assert pb[0] == 0 and pb[-1] == n and all(a <= b for a, b in zip(pb, pb[1:])) and len(pb) == plan["nparts"] + 1
    blocks = plan["row_blocks"]
    if blocks:                                   # the row blocks tile [0, rows) in order, in whole chunk-flag words
        assert blocks[0][0] == 0 and blocks[-1][1] >= rows and len(blocks) == plan["nrow_blocks"] >= 2
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:])) and all(b[0] % 1024 == 0 and b[1] > b[0] for b in blocks)
        assert plan["ncuts"] <= 32
    whole = np.zeros(n, np.int32)
    cut = np.zeros((n, max(1, len(blocks))), np.int32)
    most = 0
    for k, part in enumerate(plan["items"]):
        most = max(most, len(part))
        seen_cut = False
        for first, walked, block in part:
            assert pb[k] <= first and first + walked <= pb[k + 1]         # a partition's items stay inside its tiles
            if block is None:
                assert not seen_cut                                          # whole tiles first, row blocks last
                whole[first:first + walked] += 1
            else:
                seen_cut = True
                cut[first, block] += 1
        order = [(b, t) for t, w, b in part if b is not None]
        assert order == sorted(order)                                        # row-block-major, tiles ascending: a block's predecessor has a smaller item number
    covered_whole, covered_cut = whole == 1, (cut == 1).all(axis=1) if blocks else np.zeros(n, bool)
    assert ((covered_whole ^ covered_cut) | (n == 0)).all() and not (whole > 1).any() and not (cut > 1).any()
    assert plan["workgroups"] == -(-most // 4) * plan["nparts"]               # four items per workgroup, as many workgroups per partition as the largest needs


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("tuning", TUNINGS)
def test_every_tile_is_covered_exactly_once(shape, tuning):
    ncols, nrows, shard, shards = shape
    check(launch_plan(ncols, nrows, shard, shards, tuning=tuning))
    check(launch_plan(ncols, nrows, shard, shards, wave_slots=1000, tuning=tuning))


def test_the_rules_on_the_baseline_shapes():
    c2 = launch_plan(100_012_032, 1024)
    assert c2["nparts"] == 8 and c2["nrow_blocks"] == 1 and c2["tiles_per_group"] == 1            # whole tiles, nothing cut, nothing grouped
    c3 = launch_plan(10_002_432, 503_329)
    assert c3["ncuts"] == 0 and c3["uniform_rows"] == 16384 and c3["nrow_blocks"] == 31            # fewer tiles than wave slots: uniform blocks, every tile
    assert all(len(p) == (c3["part_begin"][k + 1] - c3["part_begin"][k]) * 31 for k, p in enumerate(c3["items"]))
    sizes = np.diff(c3["part_begin"])
    assert sizes[0] > sizes[3] * 1.15 and sizes[-1] > sizes[3] * 1.15                              # partitions of equal work: the two ends hold more (shorter) tiles
    c5 = launch_plan(100_012_032, 20_000)
    assert c5["row_blocks"] == [(0, 10240), (10240, 15360), (15360, 20000)] and c5["cut_tiles"] == 1152    # 1.5 rounds of 6144 slots over 8 partitions
    # short models (round 4): the resident-table kernel -- every wave walks a run of adjacent tiles; the runs come in rounds of as many
    # waves as the chip holds and taper (3/4 of what is left per slot), the last rounds are single tiles; the standard kernel
    # (variant 0) keeps single tiles throughout
    short = launch_plan(100_012_032, 32)
    assert short["resident_kernel"] == 1 and short["walk_slots"] == 6144 and short["walk_len"] == [5, 2, 1] and short["nrow_blocks"] == 1
    assert short["walk_base"] == [0, 5 * 6144, 7 * 6144]
    runs = short["items"][0]
    assert [w for _, w, _ in runs[:6144]] == [5] * 6144 and [w for _, w, _ in runs[6144:2 * 6144]] == [2] * 6144 and {w for _, w, _ in runs[2 * 6144:]} == {1}
    assert short["workgroups"] == 2 * 1536 + -(-(short["ntiles"] - 7 * 6144) // 4)
    assert launch_plan(100_012_032, 256)["resident_kernel"] == 1 and launch_plan(100_012_032, 257)["resident_kernel"] == 0
    few = launch_plan(SEG, 40)                         # fewer tiles than wave slots: one tile per wave, no empty workgroups but the last
    assert few["resident_kernel"] == 1 and few["workgroups"] == (few["ntiles"] + 3) // 4 and few["walk_len"] == [1]
    forced = launch_plan(100_012_032, 32, tuning=(-1, 50, -1, -1, -1, -1, -1, -1, 1))       # tests: runs of a given length throughout
    assert forced["walk_len"] == [50] and forced["workgroups"] == -(-(-(-forced["ntiles"] // 50)) // 4)
    std = launch_plan(100_012_032, 32, tuning=(-1, -1, -1, -1, -1, -1, -1, -1, 0))
    assert std["resident_kernel"] == 0 and std["tiles_per_group"] == 1 and std["nrow_blocks"] == 1
    assert launch_plan(100_012_032, 32, tuning=(-1, 4))["resident_kernel"] == 0          # a forced work distribution is the standard kernel's
    walk = launch_plan(100_012_032, 32, tuning=(-1, -4, -1, -1, -1, -1, -1, -1, 0))
    assert walk["tiles_per_group"] == 4 and walk["single_tiles"] == 768
    assert all(sum(1 for _, w, _ in p if w == 1) >= 768 and sum(1 for _, w, _ in p if w == 4) > 1000 for p in walk["items"])
    assert c2["resident_kernel"] == 0 and c5["resident_kernel"] == 0


def test_refusals():
    with pytest.raises(Exception):
        launch_plan(100, 10)                       # not whole segments
    with pytest.raises(Exception):
        launch_plan(SEG, 10, tuning=(500,))        # a row block below 1024 rows
    with pytest.raises(Exception):
        launch_plan(SEG, 10, shard_index=2, shard_count=2)
