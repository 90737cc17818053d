"""Accuracy of the bf16-split conv kernels vs float64, beside torch's own f32 convolutions."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd._lib import check, lib, ptr, stream  # noqa: E402

M = 512
g = torch.Generator().manual_seed(0)
obs = torch.randint(0, 256, (M, 4, 104, 80), dtype=torch.uint8, generator=g).cuda()
w1 = (torch.randn(16, 4, 8, 8, generator=g) * 0.06).cuda()
b1 = (torch.randn(16, generator=g) * 0.1).cuda()
w2 = (torch.randn(32, 16, 4, 4, generator=g) * 0.06).cuda()
b2 = (torch.randn(32, generator=g) * 0.1).cuda()
x64 = obs.double() / 255
y1_64 = torch.relu(F.conv2d(x64, w1.double(), b1.double(), stride=4))
y1_t = torch.relu(F.conv2d(obs.float() / 255, w1, b1, stride=4))
y1 = torch.empty(M, 475, 16, device="cuda")
check(lib.rlpyt_atari_conv1_fwd_f32(ptr(obs), None, 1, M, M, ptr(w1), ptr(b1), 1. / 255, ptr(y1), stream()))
y1_nchw = y1.reshape(M, 25, 19, 16).permute(0, 3, 1, 2)
sc = y1_64.abs().max().item()
print("conv1 fwd  max|ref| %.3f  ours %.3e  torch-f32 %.3e" % (
    sc, (y1_nchw.double() - y1_64).abs().max().item(), (y1_t.double() - y1_64).abs().max().item()))
y1_in = y1_64.float()
y2_64 = torch.relu(F.conv2d(y1_in.double(), w2.double(), b2.double(), stride=2, padding=1))
y2_t = torch.relu(F.conv2d(y1_in, w2, b2, stride=2, padding=1))
y2 = torch.empty(M, 3456, device="cuda")
y1_nhwc = y1_in.permute(0, 2, 3, 1).reshape(M, 475, 16).contiguous()
check(lib.rlpyt_atari_conv2_fwd_f32(ptr(y1_nhwc), M, ptr(w2), ptr(b2), ptr(y2), stream()))
sc = y2_64.abs().max().item()
print("conv2 fwd  max|ref| %.3f  ours %.3e  torch-f32 %.3e" % (
    sc, (y2.reshape(M, 32, 12, 9).double() - y2_64).abs().max().item(),
    (y2_t.double() - y2_64).abs().max().item()))
