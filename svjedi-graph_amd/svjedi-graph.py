#!/usr/bin/env python3
"""Drop-in for SVJedi-graph's svjedi-graph.py driver (same flags, same files, same progress lines).

Graph construction (construct-graph.py) and read mapping (minigraph) are NOT part of this package and stay
the reference's: construct-graph.py is looked up next to this script, then in $SVJEDI_GRAPH_HOME, then on PATH.
Steps 3 and 4 run the MI355X filter / genotyper that live next to this script.  With --fused the two steps
share one GPU context and the counts never leave HBM (the JSON is still written unless --no-json).
"""
import argparse
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def find_construct_graph():
    for d in (HERE, os.environ.get("SVJEDI_GRAPH_HOME", "")):
        if d and os.path.exists(os.path.join(d, "construct-graph.py")):
            return os.path.join(d, "construct-graph.py")
    w = shutil.which("construct-graph.py")
    if w:
        return w
    sys.exit("construct-graph.py not found: it is not part of this package; set SVJEDI_GRAPH_HOME to a "
             "SVJedi-graph checkout.\nExiting SVJedi-graph.")


def main():
    ap = argparse.ArgumentParser(description="Structural variations genotyping using long reads")
    ap.add_argument("-v", "--vcf", metavar="<inputVCF>", type=str, required=True, help="Set of SVs in vcf format")
    ap.add_argument("-r", "--ref", metavar="<refFA>", type=str, required=True, help="Reference genome")
    ap.add_argument("-q", "--reads", metavar="<readsFQ>", type=str, required=True, help="Reads (fastq, comma separated)")
    ap.add_argument("-p", "--prefix", metavar="<outFilesPrefix>", type=str, required=True, help="Prefix of output files")
    ap.add_argument("-t", "--threads", metavar="<threadNumber>", type=int, default=[1], help="Number of threads for mapping")
    ap.add_argument("-ms", "--minsupport", metavar="<minNbAln>", type=int, default=3,
                    help="Minimum number of alignments to genotype a SV (default: 3>=)")
    ap.add_argument("--fused", action="store_true", help="filter + genotype in one GPU context (extension)")
    ap.add_argument("--no-json", action="store_true", help="with --fused: skip writing _informative_aln.json (extension)")
    args = ap.parse_args()
    pre = args.prefix

    print("Constructing variation graph...")
    gfa = pre + ".gfa"
    p = subprocess.run("python3 {} -v {} -r {} -o {}".format(find_construct_graph(), args.vcf, args.ref, gfa), shell=True)
    if p.returncode == 1:
        sys.exit("Failed to contruct the variation graph.\nExiting SVJedi-graph.")

    print("Mapping reads on graph...")
    gaf = pre + ".gaf"
    subprocess.run(f"touch {gaf}", shell=True)
    for fq in args.reads.split(","):
        p = subprocess.run("minigraph -x lr -t{} {} {} >> {}".format(args.threads, gfa, fq, gaf), shell=True)
    if p.returncode == 1:
        sys.exit("Failed to map the reads on the graph.\nExiting SVJedi-graph.")

    out_json = pre + "_informative_aln.json"
    out_vcf = pre + "_genotype.vcf"
    if args.fused:
        print("Filtering alignment file...")
        from svjg import capi, filter as flt, genotype
        from svjg.graph import Graph
        try:
            graph = Graph.from_files(pre + "_svs_edges.json", gfa)
            ctx = capi.Context(0)
            counts, recs, data = flt.classify_file(ctx, graph, gaf, want_hits=not args.no_json)
            if not args.no_json:
                capi.write_informative_json(out_json, data, recs, graph.sv_ids)
        except Exception:
            import traceback
            traceback.print_exc()
            sys.exit("Failed to filter the alignments.\nExiting SVJedi-graph.")
        print("Genotyping SVs...")
        try:
            n = genotype.genotype_with_counts(ctx, args.vcf, graph.slot_of, out_vcf, args.minsupport)
            print("Genotyped svs: " + str(n))
        except Exception:
            import traceback
            traceback.print_exc()
            sys.exit("Failed to predict the genotypes.\nExiting SVJedi-graph.")
        return

    print("Filtering alignment file...")
    p = subprocess.run("python3 {}/filter-alignments.py -a {} -g {} -p {}".format(HERE, gaf, gfa, pre), shell=True)
    if p.returncode == 1:
        sys.exit("Failed to filter the alignments.\nExiting SVJedi-graph.")

    print("Genotyping SVs...")
    p = subprocess.run("python3 {}/predict-genotype.py -d {} -v {} --minsupport {} -o {}".format(
        HERE, out_json, args.vcf, str(args.minsupport), out_vcf), shell=True)
    if p.returncode == 1:
        sys.exit("Failed to predict the genotypes.\nExiting SVJedi-graph.")


if __name__ == "__main__":
    main()
