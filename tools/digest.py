"""One spelling of "the per-SV count vector" that does not depend on who numbered the count slots (the product's graph loader, the
oracle): the lines "sv_id\\tref\\talt\\n" of every SV with a count, sorted by sv_id, hashed.  Used by bench.py (north_star.counts_digest),
tests/c4_oracle_counts.py (tests/golden/synth/c4_oracle.json) and the tests that compare the two.  Neither product nor oracle."""
import hashlib

import numpy as np


def counts_digest(sv_ids, counts):
    c = np.asarray(counts)
    nz = np.flatnonzero(c.sum(axis=1))
    rows = sorted((str(sv_ids[i]), int(c[i, 0]), int(c[i, 1])) for i in nz.tolist())
    h = hashlib.sha256()
    for sv, a, b in rows:
        h.update(f"{sv}\t{a}\t{b}\n".encode())
    return h.hexdigest()
