"""developer tools: steady-state timing of a device call.

An MI355X that has been idle (any host sync) needs ~20 ms of work before its clocks settle (tools/sustain.py: the first 20 launches of the headline kernel
run at 276 Gsamples/s, every later group of 20 at 322-328), so a handful of launches timed one by one with a sync after each under-reports by 10-15 %.
steady() warms for at least warm_s seconds of back-to-back calls and then times at least time_s seconds of back-to-back calls between two events."""
import time
import torch


def steady(fn, warm_s=0.06, time_s=0.12, min_reps=5):
    fn()  # set-up (plans, tables, first-use allocations) is not part of the rate, and must not shorten the warm-up below
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 0
    while True:  # warm-up, also the estimate of one call's duration
        fn()
        reps += 1
        if reps % 4 == 0:
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= warm_s:
                break
    per = (time.perf_counter() - t0) / reps
    n = max(min_reps, int(time_s / max(per, 1e-6)))
    for _ in range(max(4, int(0.02 / max(per, 1e-6)))):  # the syncs above let the clocks sag: run up again, no sync from here on
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / n
