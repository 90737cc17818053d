"""Per-wave phase cycles of a persistent conv kernel (needs a -DRLPYT_TIMING build of conv.hip).
usage: phase_timing.py conv1_wgrad|conv1_fwd|conv2_fwd|conv2_bwd|conv2_bwd_x6  [n_waves]
(conv2_bwd_x6 phases: 0 gm2 staging, 1 barrier, 4 compute, 5 barrier, 6 y1 staging of the next image --
 wgrad waves 4-7 only)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd._lib import check, lib, ptr, stream  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "conv2_fwd"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M, T, B = 8192, 128, 256
obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
idx = torch.randperm(T * B, device="cuda")[:M]
dy1 = torch.randn(M, 475, 16, device="cuda")
y1 = torch.rand(M, 475, 16, device="cuda")
y2 = torch.empty(M, 3456, device="cuda")
w1, b1 = torch.randn(16, 4, 8, 8, device="cuda") * 0.05, torch.randn(16, device="cuda")
w2, b2 = torch.randn(32, 16, 4, 4, device="cuda") * 0.05, torch.randn(32, device="cuda")
ws = torch.zeros(lib.rlpyt_atari_conv_wgrad_workspace_bytes() // 4, dtype=torch.float32, device="cuda")
dw1, db1 = torch.empty(16, 4, 8, 8, device="cuda"), torch.empty(16, device="cuda")
g2, y2b = torch.randn(M, 3456, device="cuda"), torch.rand(M, 3456, device="cuda") - 0.3
dy1o = torch.empty(M, 475, 16, device="cuda")
dw2, db2 = torch.empty(32, 16, 4, 4, device="cuda"), torch.empty(32, device="cuda")
calls = {
    "conv1_wgrad": lambda: lib.rlpyt_atari_conv1_wgrad_f32(ptr(obs), ptr(idx), T, B, M, ptr(dy1), 1. / 255,
                                                           ptr(ws), ptr(dw1), ptr(db1), stream()),
    "conv1_fwd": lambda: lib.rlpyt_atari_conv1_fwd_f32(ptr(obs), ptr(idx), T, B, M, ptr(w1), ptr(b1),
                                                       1. / 255, ptr(y1), stream()),
    "conv2_fwd": lambda: lib.rlpyt_atari_conv2_fwd_f32(ptr(y1), M, ptr(w2), ptr(b2), ptr(y2), stream()),
    "conv2_bwd": lambda: lib.rlpyt_atari_conv2_bwd_f32(ptr(g2), ptr(y2b), ptr(y1), M, ptr(w2), ptr(dy1o), ptr(ws),
                                                       ptr(dw2), ptr(db2), stream()),
    "conv2_bwd_x6": lambda: lib.rlpyt_atari_conv2_bwd_x6_f32(ptr(g2), ptr(y2b), ptr(y1), M, ptr(w2), ptr(dy1o),
                                                             ptr(ws), ptr(dw2), ptr(db2), stream()),
}
for _ in range(3):
    check(calls[which]())
torch.cuda.synchronize()
buf = np.zeros(512 * 16 * 8, dtype=np.float32)
lib.rlpyt_debug_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.rlpyt_debug_timing_read(buf.ctypes.data, buf.size) == 0
rows = buf.reshape(512, 16, 8)
used = rows[:, :nw].sum(axis=(1, 2)) > 0
n_wg = int(used.sum())
per_img = rows[used][:, :nw] / (M / n_wg)
print(f"{which}: cycles per image, mean over {n_wg} workgroups:")
for w in range(nw):
    r = per_img[:, w].mean(0)
    print(f" wave {w}: " + " ".join(f"p{k}={v:6.0f}" for k, v in enumerate(r)) + f"  total={r.sum():7.0f}")
