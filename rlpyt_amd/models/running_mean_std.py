"""Running mean / variance of observations (rlpyt/models/running_mean_std.py:7-45) with
the statistics and the Chan merge computed by HIP kernels; multi-GPU: statistics are
all-reduce-averaged across ranks exactly as the reference does (one tiny RCCL all-reduce
of ``[2, *shape]`` per update)."""
import torch
import torch.distributed as dist

from .. import ops


class RunningMeanStdModel(torch.nn.Module):
    def __init__(self, shape):
        super().__init__()
        self.register_buffer("mean", torch.zeros(shape))
        self.register_buffer("var", torch.ones(shape))
        self.register_buffer("count", torch.zeros(()))
        self.shape = shape

    def update(self, x):
        batch_mean, batch_var, batch_count = ops.obs_batch_stats(x, len(self.shape))
        if dist.is_initialized():
            mean_var = torch.stack([batch_mean, batch_var])
            dist.all_reduce(mean_var)
            world_size = dist.get_world_size()
            mean_var /= world_size
            batch_count *= world_size
            batch_mean, batch_var = mean_var[0].contiguous(), mean_var[1].contiguous()
        ops.obs_rms_merge_(self.mean, self.var, self.count.view(1), batch_mean, batch_var,
                           batch_count)

    def normalize(self, x, var_clip=1e-6, obs_clip=10.):
        return ops.obs_normalize(x, self.mean, self.var, var_clip, obs_clip)
