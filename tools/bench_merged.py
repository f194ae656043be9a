"""Like-for-like rows for the reference's merged-chain table (docs/USER_API_Connecting_Blocks.md:207-222: src->mult->div->add->sink and that chain ten times over,
float and int): the run-time merged program (gr4hip_ewise_process, ONE launch) against the same blocks launched one after the other (gr4hip_math_const x 3 / x 30).
Input resident in HBM; rate = samples / median launch time (HIP events on the launch stream)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import gnuradio4_amd as G


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
    rows = []
    for dtype, value in ((torch.float32, 2.0), (torch.float32, 3.0), (torch.int32, 2)):
        x = (torch.randn(n, device="cuda") * 100).to(dtype)
        out = torch.empty_like(x)
        for reps in (1, 10):
            ops = [("Multiply", value), ("Divide", value), ("Add", -1)] * reps
            m = G.Merged(dtype, ops)
            t_merged = timed(lambda: m.process_bulk(x, out))

            def separate():
                cur = x
                for name, v in ops:
                    cur = G.math_const(name, cur, v)
            t_sep = timed(separate, reps=5, warm=1)
            es = x.element_size()
            rows.append({"chain": f"(mult->div->add)^{reps}", "dtype": str(dtype).split(".")[-1], "value": value, "n": n,
                         "merged_Gsamples_s": round(n / t_merged / 1e6, 1), "merged_TB_s": round(2 * es * n / t_merged / 1e9, 2),
                         "separate_launches_Gsamples_s": round(n / t_sep / 1e6, 1), "launches_merged": 1, "launches_separate": 3 * reps})
            print(json.dumps(rows[-1]), flush=True)
    xc = torch.randn(n // 2, dtype=torch.complex64, device="cuda")
    oc = torch.empty_like(xc)
    m = G.Merged(torch.complex64, [("Multiply", 0.5 + 0.5j), ("Rotator", 0.3, 0.0), ("Add", 1j)])
    t = timed(lambda: m.process_bulk(xc, oc))
    rows.append({"chain": "cmult->rotator->cadd", "dtype": "complex64", "n": n // 2, "merged_Gsamples_s": round(n / 2 / t / 1e6, 1), "merged_TB_s": round(16 * (n // 2) / t / 1e9, 2)})
    print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
