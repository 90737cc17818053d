"""Per-wave phase cycles of the bf16x3 conv1 kernels (needs a -DRLPYT_X3_TIMING build of conv.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd._lib import check, lib, ptr, stream  # noqa: E402

M, T, B = 8192, 128, 256
obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
idx = torch.randperm(T * B, device="cuda")[:M]
dy1 = torch.randn(M, 475, 16, device="cuda")
ws = torch.zeros(lib.rlpyt_atari_conv_wgrad_workspace_bytes() // 4, dtype=torch.float32, device="cuda")
dw1, db1 = torch.empty(16, 4, 8, 8, device="cuda"), torch.empty(16, device="cuda")
for _ in range(3):
    check(lib.rlpyt_atari_conv1_wgrad_f32(ptr(obs), ptr(idx), T, B, M, ptr(dy1), 1. / 255, ptr(ws),
                                          ptr(dw1), ptr(db1), stream()))
torch.cuda.synchronize()
PART1 = 4096 + 16
rows = ws[:512 * PART1].view(512, PART1)[256:512, :64].reshape(256, 8, 8)[:, :, :6].cpu()
names = ["barrier1", "stage_img", "stage_dy", "barrier2", "prefetch", "mfma"]
per_img = rows / 32.0
print("cycles per image, mean over 256 workgroups (one per CU):")
for w in range(8):
    r = per_img[:, w].mean(0)
    print(f" wave {w}: " + "  ".join(f"{n}={v:7.0f}" for n, v in zip(names, r.tolist())) +
          f"  total={r.sum():8.0f}")
