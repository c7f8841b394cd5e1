"""Turns a rocprofv3 run directory (rocpd sqlite) into the small text summary committed under profiles/."""
import glob
import sqlite3
import sys


def main(run_dir, out_path, title):
    lines = ["# %s" % title, "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for f in glob.glob(run_dir + "/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        for name, calls, total, avg, pct in c.execute(
                "select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append("| `%s` | %d | %.3f | %.3f | %.2f |" % (name[:90], calls, total / 1e3, avg / 1e3, pct))
        lines += ["", "dispatch details (first dispatch of each mcq kernel):", "",
                  "| kernel | grid | workgroup | LDS B | scratch B | VGPR | AGPR | SGPR |", "|---|---|---|---|---|---|---|---|"]
        seen = set()
        for r in c.execute("select name,grid_x,grid_y,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,"
                           "sgpr_count from kernels"):
            if r[0].startswith("mcq") and r[0] not in seen:
                seen.add(r[0])
                lines.append("| `%s` | %dx%d | %d | %d | %d | %d | %d | %d |" % r)
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats")
