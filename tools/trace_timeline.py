"""Timeline of a rocprofv3 --kernel-trace CSV: per launch its queue, start (us from the first launch shown), duration and the gap to
the previous launch's end on ANY queue.  usage: trace_timeline.py kernel_trace.csv [first_row [rows]]"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "frbch" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
rows = rows[first:first + n]
t0 = int(rows[0]["Start_Timestamp"])
last_end = t0
queues = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r["Queue_Id"], len(queues))
    m = re.search(r"(frbch_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
    name = (m.group(1) + (m.group(2) or "")).replace(" ", "")
    print(f"q{q}  start {(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  end {(e - t0) / 1e3:10.1f}  idle-before {(s - last_end) / 1e3:8.1f}  {name}")
    last_end = max(last_end, e)
