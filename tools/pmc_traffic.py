#!/usr/bin/env python3
"""HBM traffic of one kernel from a rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum
TCC_EA0_WRREQ_64B_sum pass (counter_collection.csv), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(read bytes = RDREQ x 64 B x 2; write bytes = WRREQ_64B x 64 + (WRREQ - WRREQ_64B) x 32), written as the JSON bench.py reads.
usage: pmc_traffic.py <counter_collection.csv> <kernel substring> <algorithmic bytes per launch> <out.json> [note]"""
import collections
import csv
import json
import sys


def main():
    path, match, algo, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
    name = None
    for r in csv.DictReader(open(path)):
        if match in r["Kernel_Name"]:
            per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            name = r["Kernel_Name"]
    if not per_dispatch:
        sys.exit(f"no dispatch of a kernel matching {match!r} in {path}")
    # the launch with the most read requests: the bench's own full launch, not a warm-up on fewer tracks
    c = max(per_dispatch.values(), key=lambda d: d.get("TCC_EA0_RDREQ_sum", 0.0))
    rd = c["TCC_EA0_RDREQ_sum"] * 64.0 * 2.0
    wr = c.get("TCC_EA0_WRREQ_64B_sum", 0.0) * 64.0 + (c.get("TCC_EA0_WRREQ_sum", 0.0) - c.get("TCC_EA0_WRREQ_64B_sum", 0.0)) * 32.0
    doc = {"source": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum; " + note,
           "kernel": name, "dispatches_seen": len(per_dispatch), "counters": dict(c),
           "correction": "read bytes = RDREQ x 64 B x 2 (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half of a 16 B/lane streaming read on "
                         "gfx950); write bytes = WRREQ_64B x 64 + (WRREQ - WRREQ_64B) x 32",
           "read_bytes": rd, "write_bytes": wr, "bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": algo,
           "ratio": (rd + wr) / algo}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("kernel", "bytes_per_launch", "algorithmic_bytes_per_launch", "ratio")}))


if __name__ == "__main__":
    main()
