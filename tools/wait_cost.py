#!/usr/bin/env python3
"""What does a cross-stream wait cost the WAITING stream when the other stream is already done?  (NOTEBOOK R6.17)"""
import torch, time
dev = torch.device("cuda:0")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
x = torch.zeros(1 << 20, device=dev)
big = torch.zeros(64 << 20, device=dev)          # 256 MB: something to leave dirty lines behind
def run(n, wait, dirty, side_busy=False):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(0.005 * 2.4e9))       # the host enqueues everything behind this
    e0.record()
    for _ in range(n):
        if dirty: big.add_(1.0)
        else: x.add_(1.0)
        if side_busy:
            with torch.cuda.stream(side): x2 = x * 2
        if wait: main.wait_stream(side)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dirty in (False, True):
    for sb in (False, True):
        a = min(run(200, False, dirty, sb) for _ in range(3)); b = min(run(200, True, dirty, sb) for _ in range(3))
        print("kernel writes %s, side stream %s: %.1f us per iteration without the wait, %.1f us with  (+%.1f us per wait)" % (
            "256 MB" if dirty else "4 MB", "busy" if sb else "idle", a, b, b - a))

# ... and when the waiting stream has to PARK: main waits for a side-stream kernel of a given length; how long after that kernel's end does main resume?
def park(us, n=20):
    res = []
    for _ in range(n):
        torch.cuda.synchronize()
        e_side, e_main = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.add_(1.0)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(us * 1e-6 * 2.4e9 / 24))      # (_sleep counts in units that run ~24x slower than the 2.4 GHz clock here: calibrated below)
            e_side.record()
        main.wait_stream(side)
        e_main.record()
        x.add_(1.0)
        torch.cuda.synchronize()
        res.append(e_side.elapsed_time(e_main) * 1e3)
    res.sort()
    return res[len(res) // 2], res[0], res[-1]
# calibrate _sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
per = e0.elapsed_time(e1) * 1e3 / 1e6
print("_sleep(1e6) = %.1f us -> %.4f us per count" % (e0.elapsed_time(e1) * 1e3, per))
def park2(us, n=20):
    res = []
    cnt = int(us / per)
    for _ in range(n):
        torch.cuda.synchronize()
        e_side, e_main = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.add_(1.0)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.cuda._sleep(cnt)
            e_side.record()
        main.wait_stream(side)
        e_main.record()
        x.add_(1.0)
        torch.cuda.synchronize()
        res.append(e_side.elapsed_time(e_main) * 1e3)
    res.sort()
    return res[len(res) // 2], res[0], res[-1]
for us in (20, 100, 300, 1000, 3000):
    print("main parked behind a %5d us side kernel: resumes %.1f us after its end (min %.1f, max %.1f)" % ((us,) + park2(us)))
