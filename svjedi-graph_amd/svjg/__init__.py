"""svjg — host side of the MI355X implementation of SVJedi-graph's filter + genotype hot path.

    svjg.graph     edge table / GFA  -> flat device tables
    svjg.capi      ctypes binding of csrc/libsvjg_hip.so (include/svjg.h)
    svjg.filter    filter-alignments.py equivalent (GAF -> counts, _informative_aln.json)
    svjg.genotype  predict-genotype.py equivalent (counts + VCF -> genotyped VCF)
    svjg.shard     byte-range sharding of a GAF over ranks + the one all-reduce of the count vector
"""
__all__ = ["graph", "capi", "filter", "genotype", "shard"]
