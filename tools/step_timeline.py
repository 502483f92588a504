"""One step of a bench.py run as a timeline, from a rocprofv3 --kernel-trace CSV: every dispatch between two consecutive
kd_turnaround_kernel launches with its queue, start offset, duration and the gap to the previous dispatch on the same queue.
    python tools/step_timeline.py <out_kernel_trace.csv> [which step, default: the middle one]"""
import csv
import re
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name)
            name = re.match(r"([A-Za-z0-9_:]+(?:<[0-9, a-z]+>)?)", name).group(1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[3].startswith("kd_turnaround_kernel")]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
    a, b = marks[k], marks[k + 1]
    t0 = rows[a][0]
    last_end = {}
    print("step %d: %.1f us from turnaround to turnaround" % (k, (rows[b][0] - t0) / 1e3))
    for s, e, q, name in rows[a:b]:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        print("q%-3s %8.1f us  +%7.1f dur  gap %6.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name))
        last_end[q] = max(e, last_end.get(q, 0))


if __name__ == "__main__":
    main()
