#!/usr/bin/env python3
"""profiles/<round>/traffic.json from the --pmc passes of tools/profile_round.sh: HBM-side bytes of k_classify_main per launch,
corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a wide coalesced
streaming read at half its bytes, so half of the text stream is added back; the 64-byte node-record fetches count as they are),
the VALU instruction count, and the clock the kernel ran at (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / kernel time)."""
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


def main():
    o = sys.argv[1]
    bench = json.load(open(f"{o}/bench_c3.json"))
    text = bench["config"]["gaf_bytes_per_gpu"]
    fetch, _ = per_launch(f"{o}/pmc_fetch_size.csv")
    write, _ = per_launch(f"{o}/pmc_write_size.csv")
    sq1, _ = per_launch(f"{o}/pmc_sq_group1.csv")
    grbm, dur = per_launch(f"{o}/pmc_grbm.csv")
    fk, wk = fetch.get("FETCH_SIZE"), write.get("WRITE_SIZE")
    out = {"_comment": __doc__.strip().replace("\n", " "), "c3": {}}
    c = out["c3"]
    c["text_bytes"] = text
    c["algorithmic_bytes"] = text + 8 * bench["config"]["count_slots"]
    if fk is not None and wk is not None:
        c["fetch_size_kib"], c["write_size_kib"] = round(fk), round(wk)
        c["traffic_bytes"] = int(fk * 1024 + text / 2 + wk * 1024)
    if "SQ_INSTS_VALU" in sq1:
        c["valu_wave_instructions"] = int(sq1["SQ_INSTS_VALU"])
    if grbm.get("GRBM_GUI_ACTIVE") and dur:
        c["clock_ghz"] = round(grbm["GRBM_GUI_ACTIVE"] / 8 / dur, 3)
    json.dump(out, open(f"{o}/traffic.json", "w"), indent=1)
    print(json.dumps(c))


if __name__ == "__main__":
    main()
