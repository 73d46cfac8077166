"""Measurement only: drain rate of a 256 x 128 fp32 accumulator tile per CU by store pattern (tools/store_probe.hip).
Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/store_probe.hip -o scratchlibs/store_probe.so"""
import ctypes, os, sys
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "scratchlibs", "store_probe.so"))
lib.store_probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
M, N = 12800, 4096
C = torch.empty(M, N, device="cuda", dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
for grid, reps in ((256, 6), (256, 25), (1600, 1), (64, 25)):
    for pat in (0, 1, 2):
        for _ in range(2):
            lib.store_probe_launch(pat, C.data_ptr(), M, N, grid, reps, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.store_probe_launch(pat, C.data_ptr(), M, N, grid, reps, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 5
        byts = grid * reps * 256 * 128 * 4
        print(f"grid {grid:5d} reps {reps:3d} pattern {pat}: {us:8.1f} us  {byts / us / 1e6:7.2f} TB/s  "
              f"{byts / us / 1e3 / min(grid, 256) / 2.1e3:6.1f} B/cycle/CU (2.1 GHz)", flush=True)
