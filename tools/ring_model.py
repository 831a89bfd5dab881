#!/usr/bin/env python3
"""The band model of the frame-parallel ring (bench.ring_model) as a table: pictures per one-GPU whole-picture step, chain (round 5) against mini-GOPs of 5
(round 6), per rank count and band size, for the configurations that have measured band tables.  No GPU needed: python tools/ring_model.py > profiles/r06_ring_model.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B          # noqa: E402


def main():
    print("# bench.ring_model: x one GPU (whole-picture step) delivered by N ranks; band tables profiles/r06_band_tables.txt (round 5: r05_band_tables.txt); chain = frame f reads f - 1, gop5 = every picture reads the")
    print("# newest multiple of 5 before it (FrameParallelRing(gop=5), bench.py --ring-gop 5).  The ceiling of the banded ring is N x whole / banded step (the banded step costs more")
    print("# than the whole-picture one: no phase planes, one stream per band); * = the band size bench.py picks")
    for name, depth, width, ctu_rows in (("4K 8-bit", 8, 3840, 34), ("4K 10-bit", 10, 3840, 34), ("8K 10-bit", 10, 7680, 68), ("1080p 8-bit", 8, 1920, 17)):
        table, whole = B.band_table(depth, width)
        kw = dict(ctu_rows=ctu_rows, depth=depth, width=width)
        print(f"\n{name}: whole-picture step {whole} ms")
        print(f"{'ranks':>5s} {'rows':>4s} {'banded ms':>9s} {'chain':>7s} {'gop5':>7s} {'ceiling':>8s}")
        for world in (2, 4, 8):
            pc, pg = B.pick_band_rows(world, **kw), B.pick_band_rows_gop(world, 5, **kw)
            for rows in sorted(r for r in table if r <= ctu_rows):
                c, g = B.ring_model(world, rows, 0, **kw) * whole, B.ring_model(world, rows, 5, **kw) * whole
                print(f"{world:5d} {rows:4d} {table[rows]:9.2f} {c:6.2f}{'*' if rows == pc else ' '} {g:6.2f}{'*' if rows == pg else ' '} {world * whole / table[rows]:8.2f}")


if __name__ == "__main__":
    main()
