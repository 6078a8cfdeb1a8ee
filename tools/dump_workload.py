"""bench.py's synthetic workloads as the two raw files `havac_benchmark --raw` takes: the 2-bit packed sequence and the int8 model.
python tools/dump_workload.py [--workload c2] [--rows N] [--columns-per-gpu N] OUT_PREFIX  -> OUT_PREFIX.seq, OUT_PREFIX.model"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--columns-per-gpu", type=int, default=0)
ap.add_argument("prefix")
args = ap.parse_args()
model, packed, ncols, _, planted = bench.make_inputs(args.workload, 1, args.rows, args.columns_per_gpu)
packed.tofile(args.prefix + ".seq")
model.tofile(args.prefix + ".model")
print(f"{args.prefix}.seq: {ncols} columns; {args.prefix}.model: {model.shape[0]} rows; {planted} planted homologs")
