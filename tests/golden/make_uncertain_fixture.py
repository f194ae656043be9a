#!/usr/bin/env python
"""Generates tests/golden/uncertain_value_ops.npz by RUNNING the reference's own meta/UncertainValue.hpp (included from /root/reference as it is by
gnuradio4_amd/host/tests/test_reference_uncertain_value.cpp, compiled against this repository's host layer): operand pairs {value, uncertainty} x 2 and the four results
a + b, a - b, a * b, a / b for UncertainValue<float> and UncertainValue<double>.  Data only -- inputs and the reference's outputs; run in the container (the reference
does not travel to the GPU box, the fixture does)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/meta/include/gnuradio-4.0/meta/UncertainValue.hpp"


def table(reference=True):
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "uv")
        cmd = ["g++", "-std=c++20", "-w", "-O1", "-I" + os.path.join(ROOT, "gnuradio4_amd", "host", "include")]
        if reference:
            cmd.append('-DREF_UNCERTAIN_HPP="%s"' % REF)
        cmd += [os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_uncertain_value.cpp"), "-o", exe]
        subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
        out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    rows = {"f32": [], "f64": []}
    for ln in out.splitlines():
        p = ln.split()
        if p and p[0] in rows:
            rows[p[0]].append([float(v) for v in p[1:]])
    assert ("reference UncertainValue.hpp unmodified: done" if reference else "host layer gr::UncertainValue: done") in out
    return {k: np.array(v) for k, v in rows.items()}


if __name__ == "__main__":
    t = table(True)
    np.savez(os.path.join(ROOT, "tests", "golden", "uncertain_value_ops.npz"), f32=t["f32"].astype(np.float32), f64=t["f64"],
             columns=np.array("a.value a.uncertainty b.value b.uncertainty | (a+b).value .uncertainty (a-b) (a*b) (a/b)"))
    print({k: v.shape for k, v in t.items()})
