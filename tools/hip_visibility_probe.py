import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
import torch
if mode in ("avail", "avail_sub"):
    print("avail", torch.cuda.is_available(), flush=True)
if mode == "avail_sub":
    import subprocess; subprocess.check_call(["true"])
if mode == "count":
    print("count", torch.cuda.device_count(), flush=True)
import ctypes as C
from rtlsdr_amd import capi
lib = capi.load()
h = C.c_void_p(); cfg = capi.RtlfmCfg.default()
r = lib.rtlfm_gpu_create(C.byref(cfg), 1, 0, C.byref(h))
print(mode, "create ->", r, flush=True)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "libhsa-runtime" in l]
print(sorted(set(maps)))
