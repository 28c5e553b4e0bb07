"""Shared by the examples: data that ships with the repository (tests/golden/*.npz: the reference's hand mesh, depth image and
photographs, stored there by tests/golden/make_golden.py), a timer, an optional PNG / NPZ dump."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def hand_mesh():
    d = golden("hand_mesh.npz")
    return d["vertices"], d["faces"].astype(np.int64)


def run(step, iterations, report_every, label):
    """`step()` -> energy (a device tensor): iterate without a host synchronisation, print a few energies at the end"""
    import torch

    energies = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iterations):
        energies.append(step().clone())  # (the returned tensor is a buffer of the fitter: overwritten by the next step)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iterations
    values = torch.stack(energies).cpu().numpy()
    for i in range(0, iterations, report_every):
        print(f"{label}: iteration {i:4d}  energy {values[i]:.6f}")
    print(f"{label}: iteration {iterations - 1:4d}  energy {values[-1]:.6f}   ({dt * 1e3:.3f} ms per iteration)")
    return values
