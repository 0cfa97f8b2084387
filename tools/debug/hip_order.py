"""Which HIP runtime(s) end up in the process when libsphhip.so is loaded before / after torch, and whether a
HIP call from libsphhip.so sees the GPU in both orders."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1]


def maps():
    seen = set()
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line or "libhsa-runtime64" in line:
            seen.add(line.split()[-1])
    return sorted(seen)


if order == "lib_first":
    from sph_amd import lib
    L = lib.load()
    import torch
else:
    import torch
    from sph_amd import lib
    L = lib.load()
x = torch.zeros(4, device="cuda")
torch.cuda.synchronize()
print(order, maps())
hip = ctypes.CDLL("libamdhip64.so.7")
dev = ctypes.c_int(-1)
print("hipGetDevice ->", hip.hipGetDevice(ctypes.byref(dev)), dev.value)
cnt = ctypes.c_int(-1)
print("hipGetDeviceCount ->", hip.hipGetDeviceCount(ctypes.byref(cnt)), cnt.value)
print("hipGetDevice ->", hip.hipGetDevice(ctypes.byref(dev)), dev.value)
