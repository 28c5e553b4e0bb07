"""Timeline of the library's kernels from a rocprofv3 --kernel-trace csv (two-stream experiments): which kernels overlap.
    python tools/trace_timeline.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    short = next((s for s in ("setup_bin", "tile_scan", "raster_fwd_fast", "finalize", "raster_bwd", "wait_flag", "views_gradient_sum") if s in name), None)
    if short is None:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r["Queue_Id"], r.get("Stream_Id", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
tail = rows[-n:]
t0 = tail[0][0]
print("  start     end     dur  queue/stream kernel")
for s, e, k, q, st in tail:
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f}  {q:>3}/{st:<3} {k}")
# occupancy classes over the window: a forward raster running / only latency-bound kernels / idle
ev = []
for s, e, k, q, st in rows[len(rows) // 2 :]:
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
live = {}
acc = {"forward (+ anything)": 0, "two forwards": 0, "set-up / finalize / scan only": 0, "idle": 0}
last = ev[0][0]
for t, d, k in ev:
    dt = t - last
    nf = live.get("raster_fwd_fast", 0)
    other = sum(v for kk, v in live.items() if kk != "raster_fwd_fast")
    if nf >= 2:
        acc["two forwards"] += dt
    if nf >= 1:
        acc["forward (+ anything)"] += dt
    elif other:
        acc["set-up / finalize / scan only"] += dt
    else:
        acc["idle"] += dt
    live[k] = live.get(k, 0) + d
    last = t
span = ev[-1][0] - ev[0][0]
print({k: round(v / span, 3) for k, v in acc.items()}, f"window {span / 1e3:.0f} us")
