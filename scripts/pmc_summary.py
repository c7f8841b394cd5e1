"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv output) and a --kernel-trace --stats pass into profiles/.

Units / corrections (per /opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB
(hbm_bytes = counter * 1024); on gfx950 FETCH_SIZE reports HALF of the bytes of a wide coalesced streaming read
(128-byte requests tallied at 64 B) -> doubled here; the correction is calibrated for 16-B/lane streams only and our
kernels mostly issue 8-B/lane loads, so `traffic` is an estimate (ratios between variants of one kernel are unaffected).
"""
import collections
import csv
import glob
import json
import sys


def counter_means(run_dir, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(run_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("mcq"):
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main(stats_dir, fetch_dir, write_dir, out_prefix, title):
    fetch = counter_means(fetch_dir, "FETCH_SIZE")
    write = counter_means(write_dir, "WRITE_SIZE")
    stats = {}
    for f in glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Name"].startswith("mcq"):
                stats[r["Name"].split("(")[0]] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6,
                                                      pct=float(r["Percentage"]))
    out = {"title": title, "kernels": {}}
    lines = ["# " + title, "", "| kernel | calls | avg ms | % | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic GB/launch (2*fetch+write) |",
             "|---|---|---|---|---|---|---|"]
    for k in sorted(stats, key=lambda k: -stats[k]["pct"]):
        fe, wr = fetch.get(k, float("nan")), write.get(k, float("nan"))
        tr = (2.0 * fe + wr) * 1024.0
        out["kernels"][k] = dict(stats[k], fetch_kib=fe, write_kib=wr, traffic_bytes=tr)
        lines.append("| `%s` | %d | %.3f | %.2f | %.4g | %.4g | %.1f |" % (k, stats[k]["calls"], stats[k]["avg_ms"], stats[k]["pct"],
                                                                         fe, wr, tr / 1e9))
    json.dump(out, open(out_prefix + ".json", "w"), indent=1)
    open(out_prefix + ".md", "w").write("\n".join(lines) + "\n\n" + __doc__)
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:6])
