"""gpurun_out/<dir>/tl/**/*kernel_trace.csv (rocprofv3 --kernel-trace over tests/tools/lanes_timeline.py) -> a text timeline of the last
frames: per kernel launch its queue, start and end (us, relative), and how much of the span had >= 2 / >= 3 kernels in flight.
    python profiles/summarize_timeline.py gpurun_out/<dir>/tl [out.md]"""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
tail = [r for r in rows if "k_bounce" in r[2] or "k_resolve" in r[2] or "k_post" in r[2] or "k_bloom" in r[2]]
tail = tail[-160:]
t0 = tail[0][0]
out = ["# kernel timeline of the last asynchronous frames (us since the first row; queue = HW queue of the launch)", "", "| start | end | dur | queue | kernel |", "|---|---|---|---|---|"]
for s, e, n, q, st in tail[:80]:
    out.append("| %.1f | %.1f | %.1f | %s | %s |" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
ev = sorted([(s, 1) for s, e, *_ in tail] + [(e, -1) for s, e, *_ in tail])
cur, last, acc = 0, ev[0][0], {}
for t, dlt in ev:
    acc[cur] = acc.get(cur, 0) + (t - last); last = t; cur += dlt
span = tail[-1][1] - tail[0][0]
frames = sum(1 for r in tail if "k_resolve" in r[2])
out += ["", "span %.1f us, %d frames resolved in it: %.1f us per frame; kernels in flight: " % (span / 1e3, frames, span / 1e3 / max(frames, 1)) +
        ", ".join("%d: %.0f %%" % (k, 100 * v / span) for k, v in sorted(acc.items())),
        "sum of kernel durations / span = %.2f" % (sum(e - s for s, e, *_ in tail) / span)]
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
