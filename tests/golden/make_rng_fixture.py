"""Generates tests/golden/rng_seed42.npz from oracle/_ref/libgr4ref.so, i.e. from the REFERENCE'S OWN rng code
compiled where it lies under /root/reference (container only).  The fixture is data; this script is its provenance."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O  # noqa: E402

O.build()
R = O.ref_lib()
assert R is not None, "oracle/_ref/libgr4ref.so missing: run `make -C oracle ref` inside the authoring container"
d = np.empty(64, np.uint64)
R.gr4ref_xoshiro_draws(42, d.ctypes.data, 64)
f = np.empty(256, np.float32)
R.gr4ref_gauss_fill_f32(42, f.ctypes.data, 256, 1.0, 0.0)
c = np.empty(256, np.complex64)
R.gr4ref_gauss_fill_c32(42, c.ctypes.data, 256, 1.0, 0.0)
np.savez(os.path.join(os.path.dirname(__file__), "rng_seed42.npz"), draws=d, gauss_f32=f, gauss_c32=c)
print("wrote rng_seed42.npz")
