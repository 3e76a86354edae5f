#!/usr/bin/env python3
"""Drop-in for SVJedi-graph's predict-genotype.py (same flags, same files) running on an MI355X.

    predict-genotype.py -d P_informative_aln.json -v VCF --minsupport N -o P_genotype.vcf   (svjedi-graph.py:124)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser(description="Structural variations genotyping using long reads")
    ap.add_argument("-d", "--aln", metavar="<alndict>", nargs=1, required=True)
    ap.add_argument("-v", "--vcf", metavar="<vcffile>", help="vcf format", required=True)
    ap.add_argument("-o", "--output", metavar="<output>", nargs=1, help="output file")
    ap.add_argument("-e", "--err", nargs=1, type=float, help="allele error probability")
    ap.add_argument("-ms", "--minsupport", metavar="<minNbAln>", type=int, default=3,
                    help="Minimum number of alignments to genotype a SV (default: 3>=)")
    args = ap.parse_args()
    out = "genotype_results.txt" if args.output is None else args.output[0]
    err = args.err[0] if args.err is not None else 0.00005
    from svjg import genotype
    genotype.run(args.aln[0], args.vcf, out, args.minsupport, err)


if __name__ == "__main__":
    main()
