"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a markdown table: per-kernel calls,
total / average duration (microseconds), share.
usage: python tools/rocpd_summary.py results.db [steps] > profiles/x.md"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print("| kernel | calls | total us | avg us | % |" + (" us per step |" if steps else ""))
    print("|---|---|---|---|---|" + ("---|" if steps else ""))
    for name, calls, total, avg, pct in rows[:70]:
        line = f"| `{name[:120]}` | {calls} | {total:.1f} | {avg:.3f} | {pct:.2f} |"
        if steps:
            line += f" {total / steps:.1f} |"
        print(line)
    print(f"\ntotal kernel time: {tot:.1f} us" + (f" = {tot / steps:.1f} us per step over {steps} steps" if steps else ""))


if __name__ == "__main__":
    main()
