#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_passes.sh.

usage: python tools/pmc_summary.py gpurun_out/pmc_<tag> [out.csv]
FETCH_SIZE is reported as collected (KB) and corrected (x2, MI355X_MICROARCH.md: gfx950 tallies 128-B
requests at 64 B); WRITE_SIZE as collected."""
import csv, glob, os, re, sys, collections

def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")

def rows_of(f):
    """(dispatch id, kernel name, counter, value) rows of a rocprofv3 counter file (csv or rocpd sqlite)"""
    if f.endswith(".db"):
        import sqlite3
        con = sqlite3.connect(f)
        yield from con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection")
        con.close()
    else:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"]


def main():
    d = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    files = glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True)
    files += glob.glob(os.path.join(d, "*", "**", "*_results.db"), recursive=True)
    for f in files:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for disp, kname, ctr, val in rows_of(f):
            k = short(kname)
            if k is None:
                continue
            per_dispatch[(disp, ctr)] += float(val)
            names[disp] = k
        for (disp, ctr), v in per_dispatch.items():
            a = acc[names[disp]][ctr]
            a[0] += v; a[1] += 1
    ctrs = sorted({c for k in acc for c in acc[k]})
    rows = []
    for k in sorted(acc):
        row = {"kernel": k, "launches": max(a[1] for a in acc[k].values())}
        for c in ctrs:
            if c in acc[k]:
                row[c] = acc[k][c][0] / acc[k][c][1]
        if "FETCH_SIZE" in row:
            row["hbm_read_MB_corrected"] = row["FETCH_SIZE"] * 2 * 1024 / 1e6
        if "WRITE_SIZE" in row:
            row["hbm_write_MB"] = row["WRITE_SIZE"] * 1024 / 1e6
        rows.append(row)
    cols = ["kernel", "launches", "hbm_read_MB_corrected", "hbm_write_MB"] + ctrs
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    wr = csv.DictWriter(out, fieldnames=cols, extrasaction="ignore")
    wr.writeheader()
    for r in rows:
        wr.writerow({c: (("%.4g" % r[c]) if isinstance(r.get(c), float) else r.get(c, "")) for c in cols})

if __name__ == "__main__":
    main()
