#!/usr/bin/env python3
"""Per-stream timeline of the big kernels of the last bench step from a rocprofv3 --kernel-trace sqlite db."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name, start, end, stream_id, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
loss = [i for i, r in enumerate(rows) if "va_loss_grad" in r[0] or "va_loss_kernel" in r[0] or "va_loss_fused" in r[0]]
# one step = from a loss-gradient kernel to the next; of the traced steps take the one with the shortest span (a profiler flush or
# any other host stall inside a step shows up as milliseconds of idle GPU and is not what the step costs)
cands = [rows[a:b] for a, b in zip(loss[:-1], loss[1:])][-6:]
seg = min(cands, key=lambda sg: sg[-1][2] - sg[0][1])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |m3t_gru::", "", n)
    return n[:44]


base = seg[0][1]
out = []
for n, s, e, st, gx, gy, gz, wx in seg:
    k = (short(n) + " g%dx%dx%d" % (gx // max(1, wx), gy, gz), st)
    if out and out[-1][0] == k and s - out[-1][2] < 20000:
        out[-1][2] = e; out[-1][3] += 1; out[-1][4] += e - s
    else:
        out.append([k, s, e, 1, e - s])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 150e3
lo, hi = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (-1e30, 1e30)      # optional window in ms
for k, s, e, c, busy in out:
    if busy > thr and lo <= (s - base) / 1e6 <= hi:
        print("st%d  %8.3f -> %8.3f ms  n=%4d busy %7.3f  %s" % (k[1], (s - base) / 1e6, (e - base) / 1e6, c, busy / 1e6, k[0]))
print("step span %.3f ms" % ((seg[-1][2] - base) / 1e6))
# GPU idle inside the step: time covered by no kernel on any stream, and the largest holes (with what ran before / after)
iv = sorted((r_[1], r_[2], short(r_[0])) for r_ in seg)
idle, holes, cur_end, prev = 0, [], iv[0][1], iv[0][2]
for s_, e_, n_ in iv[1:]:
    if s_ > cur_end:
        idle += s_ - cur_end
        holes.append((s_ - cur_end, cur_end, prev, n_))
    if e_ > cur_end:
        cur_end, prev = e_, n_
print("GPU idle (no kernel on any stream) %.3f ms in %d holes; largest:" % (idle / 1e6, len(holes)))
for d, at, a, b in sorted(holes, reverse=True)[:12]:
    print("  %6.1f us at %7.3f ms  after %-40s before %s" % (d / 1e3, (at - base) / 1e6, a[:40], b[:40]))
