"""How long the host takes to see the end of a stream: torch.cuda.synchronize() behind one tiny kernel, with the device's default scheduling flag and with
hipDeviceScheduleSpin (hipSetDeviceFlags before the first use of the device).  GPU box:  python tools/sync_spin_probe.py [spin]"""
import ctypes
import sys
import time

spin = "spin" in sys.argv
hip = ctypes.CDLL("libamdhip64.so")
rc = hip.hipSetDeviceFlags(ctypes.c_uint(1)) if spin else 0  # hipDeviceScheduleSpin = 0x1
import torch  # noqa: E402

x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
for _ in range(50):
    x.add_(1.0)
    torch.cuda.synchronize()
lat = []
for _ in range(300):
    t0 = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
lat.sort()
print(f"{'spin' if spin else 'default'} (hipSetDeviceFlags rc {rc}): launch + synchronize of one tiny kernel: median {lat[150] * 1e6:.1f} us, p10 {lat[30] * 1e6:.1f}, p90 {lat[270] * 1e6:.1f}")
