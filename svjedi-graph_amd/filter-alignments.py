#!/usr/bin/env python3
"""Drop-in for SVJedi-graph's filter-alignments.py (same flags, same files) running on an MI355X.

    filter-alignments.py -a P.gaf -g P.gfa -p P        (svjedi-graph.py:114)
reads  P_svs_edges.json, writes P_informative_aln.json.  `-a -` reads the GAF from standard input and classifies it while it
arrives (minigraph ... | filter-alignments.py -a - ...).  Any input the reference would die on makes this
script exit with code 1 as well (uncaught exception), which is what svjedi-graph.py:117 tests for.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser(description="---")
    ap.add_argument("-a", "--gaf", metavar="<align_file>", nargs=1, help="align file in gaf format", required=True)
    ap.add_argument("-g", "--gfa", metavar="<graph_file>", nargs=1, help="variant graph in gfa format", required=True)
    ap.add_argument("-i", "--gfainfo", metavar="<gfa_info>", nargs=1, help="gfa info", required=False)
    ap.add_argument("-O", "--dover", metavar="<min_breakpoint_overlap>", nargs=1, required=False, default=100)
    ap.add_argument("-o", "--outputDir", metavar="<outputDirectory>", type=str, required=False)
    ap.add_argument("-p", "--prefix", metavar="<prefix", type=str, required=False)
    args = ap.parse_args()
    from svjg import filter as flt
    # -O: argparse (nargs=1, no type) hands the reference a list of one string; it dies with TypeError where it first compares an
    # overlap with it (filter-alignments.py:269) — at the first link with a candidate SV, not at start-up
    flt.run(args.gaf[0], args.gfa[0], args.prefix, args.outputDir, dover_given=args.dover != 100)


if __name__ == "__main__":
    main()
