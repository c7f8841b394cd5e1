"""Summarises the rocprofv3 passes of scripts/profile_round.sh (csv output) into profiles/<tag>_pmc.{json,md} and
profiles/latest_pmc.json (what bench.py quotes as `roofline.traffic`, with its source).

  python scripts/pmc_summary.py gpurun_out <tag> profiles/<tag>_pmc "<title>"

HBM traffic -- how the bytes are counted (MI355X_MICROARCH.md, section HBM, and this round's calibration):
  * reads : sum over the L2's memory-side read requests BY SIZE CLASS, 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B
            (TCC_EA0_RDREQ_*_sum), when the 64-B / 128-B class counters are populated on this box; otherwise FETCH_SIZE (KiB)
            times the correction factor measured on scripts/calib_stream (a stream of known size with 8-byte and with 16-byte loads
            per lane: the gfx950 FETCH_SIZE formula tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md);
  * writes: WRITE_SIZE (KiB) = 32 * (WRREQ - WRREQ_64B) + 64 * WRREQ_64B, times the factor measured on the calibration stream.
The calibration rows and the factors used are part of the output.  Counter passes are separate runs of the same command (a
counter set per run, never mixed with trace domains); per-kernel values are means over the dispatches of a run.
"""
import collections
import csv
import glob
import json
import os
import sys


def counters(run_dir, prefix=("mcq", "calib")):
    """{kernel: {counter: mean value over dispatches}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(run_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith(prefix):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def merged(base, tag, kind, sets):
    out = collections.defaultdict(dict)
    for s in sets:
        for k, d in counters(os.path.join(base, "%s_%s_%s" % (tag, kind, s))).items():
            out[k].update(d)
    return out


def read_bytes(c, fetch_factor):
    n32, n64, n128, tot = (c.get("TCC_EA0_RDREQ_32B_sum"), c.get("TCC_EA0_RDREQ_64B_sum"), c.get("TCC_EA0_RDREQ_128B_sum"),
                           c.get("TCC_EA0_RDREQ_sum"))
    if None not in (n32, n64, n128, tot) and tot > 0 and abs((n32 + n64 + n128) - tot) <= 0.02 * tot:
        return 32.0 * n32 + 64.0 * n64 + 128.0 * n128, "request size classes"
    if "FETCH_SIZE" in c:
        return c["FETCH_SIZE"] * 1024.0 * fetch_factor, "FETCH_SIZE x %.3f" % fetch_factor
    return float("nan"), "n/a"


def main(base, tag, out_prefix, title):
    stats = {}
    for f in glob.glob(os.path.join(base, tag + "_stats") + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Name"].startswith("mcq"):
                stats[r["Name"].split("(")[0]] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, pct=float(r["Percentage"]))
    pmc = merged(base, tag, "pmc", ("wave", "mfma", "lds", "fetch", "write", "rdreq", "l2"))
    cal = merged(base, tag, "calib", ("fetch", "write", "rdreq"))
    known = None
    for f in glob.glob(os.path.join(base, tag + "_calib_*.log")):
        for line in open(f):
            if "bytes_per_kernel" in line:
                known = float(json.loads(line)["bytes_per_kernel"])
    # ---- calibration: counter bytes / known bytes for the three stream shapes
    calib = {"known_bytes_per_kernel": known, "rows": {}}
    f8 = f16 = w8 = None
    if known:
        for k, c in cal.items():
            row = dict(c)
            if "FETCH_SIZE" in c:
                row["FETCH_SIZE_bytes_over_known"] = c["FETCH_SIZE"] * 1024.0 / known
            if "WRITE_SIZE" in c:
                row["WRITE_SIZE_bytes_over_known"] = c["WRITE_SIZE"] * 1024.0 / known
            rb, how = read_bytes(c, 1.0)
            if how == "request size classes":
                row["size_class_read_bytes_over_known"] = rb / known
            calib["rows"][k] = row
        f8 = calib["rows"].get("calib_read8", {}).get("FETCH_SIZE_bytes_over_known")
        f16 = calib["rows"].get("calib_read16", {}).get("FETCH_SIZE_bytes_over_known")
        w8 = calib["rows"].get("calib_write8", {}).get("WRITE_SIZE_bytes_over_known")
    # mcq_solve_kernel reads mostly 16-byte-per-lane rows in the sweeps and 8-byte-per-lane rows elsewhere; the two factors are
    # reported, the 16-byte one is applied when they differ (the sweeps + factorisation H fetch dominate the bytes)
    fetch_factor = 1.0 / f16 if f16 else (1.0 / f8 if f8 else 2.0)
    write_factor = 1.0 / w8 if w8 else 1.0
    calib.update(fetch_factor_used=fetch_factor, write_factor_used=write_factor,
                 fetch_factor_8B_per_lane=(1.0 / f8 if f8 else None), fetch_factor_16B_per_lane=(1.0 / f16 if f16 else None))
    sha = None
    sha_file = os.path.join(base, tag + "_source_sha.txt")
    if os.path.exists(sha_file):
        sha = open(sha_file).read().strip() or None
    out = {"title": title, "source": "gpurun_out/%s_* (scripts/profile_round.sh)" % tag, "engine_source_sha256": sha,
           "calibration": calib, "kernels": {}}
    lines = ["# " + title, "",
             "| kernel | calls | avg ms | % | read GB/launch | how | write GB/launch | traffic GB/launch | L2 hit % |", "|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(stats, key=lambda k: -stats[k]["pct"]):
        c = pmc.get(k, {})
        rb, how = read_bytes(c, fetch_factor)
        wb = c.get("WRITE_SIZE", float("nan")) * 1024.0 * write_factor
        hit, miss = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
        l2 = 100.0 * hit / (hit + miss) if hit is not None and miss is not None and hit + miss > 0 else float("nan")
        out["kernels"][k] = dict(stats[k], read_bytes=rb, read_bytes_how=how, write_bytes=wb, traffic_bytes=rb + wb, l2_hit_pct=l2,
                                 counters=c)
        lines.append("| `%s` | %d | %.3f | %.2f | %.1f | %s | %.1f | %.1f | %.1f |" % (k, stats[k]["calls"], stats[k]["avg_ms"], stats[k]["pct"],
                                                                                      rb / 1e9, how, wb / 1e9, (rb + wb) / 1e9, l2))
    # ---- the solver kernel's issue / stall / matrix-core picture
    c = pmc.get("mcq_solve_kernel", {})
    if c:
        lines += ["", "## mcq_solve_kernel: SQ counters (means per dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, "
                  "SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES in cycles)", "", "| counter | value |", "|---|---|"]
        for name in sorted(c):
            lines.append("| %s | %.6g |" % (name, c[name]))
        wc = c.get("SQ_WAVE_CYCLES")
        d = {}
        if wc:
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                       "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if nm in c:
                    d[nm + "_over_WAVE_CYCLES"] = c[nm] / wc
        gui = c.get("GRBM_GUI_ACTIVE")
        ms_k = stats.get("mcq_solve_kernel", {}).get("avg_ms")
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # MfmaUtil as rocprofv3 derives it: busy cycles / (GUI-active cycles x SIMDs), 256 CUs x 4 SIMDs.  The per-dispatch
            # GRBM_GUI_ACTIVE value in the csv is the SUM over the 8 XCDs (rocprofv3's own formula takes the max over them).
            d["gui_active_cycles_per_xcd"] = gui / 8.0
            d["MfmaUtil_pct"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0)
            if ms_k:
                d["effective_clock_ghz"] = gui / 8.0 / (ms_k * 1e-3) / 1e9
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
            d["mfma_f64_flops_per_launch"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
        if "SQ_INSTS_VALU_FMA_F64" in c:
            d["valu_fma_f64_flops_per_launch"] = c["SQ_INSTS_VALU_FMA_F64"] * 64.0 * 2.0
        ms = stats.get("mcq_solve_kernel", {}).get("avg_ms")
        if ms:
            fl = d.get("mfma_f64_flops_per_launch", 0.0) + d.get("valu_fma_f64_flops_per_launch", 0.0)
            d["fp64_tflops_counted"] = fl / (ms * 1e-3) / 1e12
            d["fp64_frac_of_78.6_tflops"] = d["fp64_tflops_counted"] / 78.6
        out["kernels"]["mcq_solve_kernel"]["derived"] = d
        lines += ["", "| derived | value |", "|---|---|"] + ["| %s | %.5g |" % kv for kv in sorted(d.items())]
    lines += ["", "## calibration (scripts/calib_stream: streams of %s bytes)" % (("%.0f" % known) if known else "?"), "",
              "```", json.dumps(calib, indent=1), "```", "", __doc__]
    json.dump(out, open(out_prefix + ".json", "w"), indent=1)
    json.dump(out, open(os.path.join(os.path.dirname(out_prefix), "latest_pmc.json"), "w"), indent=1)
    open(out_prefix + ".md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
