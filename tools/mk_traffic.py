#!/usr/bin/env python3
"""profiles/<round>/traffic.json from the --pmc passes of tools/profile_round.sh, one entry per workload: HBM-side bytes of
k_classify_main per launch, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies a wide coalesced streaming read at half its bytes, so half of the text stream is added back; the 64-byte node-record
fetches count as they are), the VALU instruction count, L2 requests / hits / misses, and the clock the kernel ran at
(GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / kernel time)."""
import csv
import json
import sys
from collections import defaultdict


def per_launch(path):
    acc, dur = defaultdict(list), []
    try:
        for r in csv.DictReader(open(path)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    except OSError:
        pass
    return {k: sum(v) / len(v) for k, v in acc.items()}, (sum(dur) / len(dur) if dur else None)


def entry(o, w):
    bench = json.load(open(f"{o}/bench_{w}.json"))
    text = bench["config"]["gaf_bytes_per_gpu"]
    fetch, _ = per_launch(f"{o}/pmc_fetch_size_{w}.csv")
    write, _ = per_launch(f"{o}/pmc_write_size_{w}.csv")
    sq1, _ = per_launch(f"{o}/pmc_sq_group1_{w}.csv")
    tcc, _ = per_launch(f"{o}/pmc_tcc_{w}.csv")
    grbm, dur = per_launch(f"{o}/pmc_grbm_{w}.csv")
    fk, wk = fetch.get("FETCH_SIZE"), write.get("WRITE_SIZE")
    c = {"text_bytes": text, "algorithmic_bytes": text + 8 * bench["config"]["count_slots"]}
    if fk is not None and wk is not None:
        c["fetch_size_kib"], c["write_size_kib"] = round(fk), round(wk)
        c["traffic_bytes"] = int(fk * 1024 + text / 2 + wk * 1024)
    if "SQ_INSTS_VALU" in sq1:
        c["valu_wave_instructions"] = int(sq1["SQ_INSTS_VALU"])
    for k in ("TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
        if k in tcc:
            c[k.lower()] = int(tcc[k])
    if grbm.get("GRBM_GUI_ACTIVE") and dur:
        c["clock_ghz"] = round(grbm["GRBM_GUI_ACTIVE"] / 8 / dur, 3)
    return c


def main():
    o = sys.argv[1]
    out = {"_comment": __doc__.strip().replace("\n", " ")}
    for w in sys.argv[2:] or ["c3"]:
        try:
            out[w] = entry(o, w)
        except (OSError, KeyError, ValueError) as e:
            print(f"{w}: {e}", file=sys.stderr)
    json.dump(out, open(f"{o}/traffic.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
