"""Per-wave phase cycles of conv2_bwd_kernel (needs a -DRLPYT_B2_TIMING build of conv.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd._lib import check, lib, ptr, stream  # noqa: E402

M = 8192
y1 = torch.rand((M, 475, 16), device="cuda") - 0.3
y2 = torch.rand((M, 3456), device="cuda") - 0.3
g2 = torch.randn(M, 3456, device="cuda")
w2 = torch.randn(32, 16, 4, 4, device="cuda") * 0.1
dy1 = torch.empty_like(y1)
ws = torch.zeros(lib.rlpyt_atari_conv_wgrad_workspace_bytes() // 4, dtype=torch.float32, device="cuda")
dw2, db2 = torch.empty_like(w2), torch.empty(32, device="cuda")
for _ in range(3):
    check(lib.rlpyt_atari_conv2_bwd_f32(ptr(g2), ptr(y2), ptr(y1), M, ptr(w2), ptr(dy1), ptr(ws),
                                        ptr(dw2), ptr(db2), stream()))
torch.cuda.synchronize()
PART2 = 8192 + 32
rows = ws.view(512, PART2)[256:512, :64].reshape(256, 8, 8)[:, :, :6].cpu()   # [wg, wave, phase]
names = ["prefetch_issue", "compute_A", "stage", "compute_B", "barrier", "-"]
per_img = rows / 32.0
print("cycles per image, mean over 256 workgroups (one per CU):")
for w in range(8):
    r = per_img[:, w].mean(0)
    print(f" wave {w} ({'dgrad' if w < 4 else 'wgrad'} q={w & 3}): " +
          "  ".join(f"{n}={v:8.0f}" for n, v in zip(names[:5], r[:5].tolist())) +
          f"  total={r[:5].sum():8.0f}")
