"""What every algorithm on this path shares.

The runner-facing contract is the reference's (``rlpyt/algos/base.py:3-68``: ``initialize`` ->
``optimize_agent(itr, samples) -> OptInfo``, ``optim_state_dict``, ``batch_size``, the class
attributes the runner reads).  On top of it this base owns the two things that differ from a CPU
algorithm here: sample batches arrive as HBM tensors (anything that is not already on the agent's
device is moved once, asynchronously) and diagnostics leave the device once per call instead of
once per minibatch."""
import torch


class RlAlgorithm:
    # read by the runner (minibatch_rl.py:78,105,124,179-189)
    opt_info_fields = ()
    bootstrap_value = False
    update_counter = 0
    optimizer = None
    _batch_size = None

    # ------------------------------------------------------------------ runner contract
    def initialize(self, agent, n_itr, batch_spec, mid_batch_reset, examples=None,
                   world_size=1, rank=0):
        raise NotImplementedError(f"{type(self).__name__}.initialize")

    def optimize_agent(self, itr, samples=None, sampler_itr=None):
        raise NotImplementedError(f"{type(self).__name__}.optimize_agent")

    @property
    def batch_size(self):
        """Training batch size per update (the runner's replay-ratio bookkeeping)."""
        return self._batch_size

    def optim_state_dict(self):
        return None if self.optimizer is None else self.optimizer.state_dict()

    def load_optim_state_dict(self, state_dict):
        self.optimizer.load_state_dict(state_dict)

    # ------------------------------------------------------------------ optimizer
    def make_optimizer(self, params, OptimCls, lr, optim_kwargs=None):
        """``OptimCls(params, lr=lr, **optim_kwargs)`` as the reference builds it
        (rlpyt/algos/pg/base.py:34-37, dqn/dqn.py:108-112) -- except that plain
        ``torch.optim.Adam`` over float32 device parameters becomes ``ClipAdam`` (same class
        hierarchy, hyper-parameters and state_dict; clip + step in two launches)."""
        optim_kwargs = dict(optim_kwargs or {})
        params = list(params)
        import os
        if (os.environ.get("RLPYT_CLIP_ADAM", "1") != "0" and OptimCls is torch.optim.Adam and params and all(
                p.is_cuda and p.dtype == torch.float32 for p in params)
                and not any(optim_kwargs.get(k) for k in ("amsgrad", "maximize", "capturable",
                                                          "differentiable"))):
            from ..optim import ClipAdam
            return ClipAdam(params, lr=lr, **optim_kwargs)
        if (OptimCls in (torch.optim.Adam, torch.optim.AdamW) and "fused" not in optim_kwargs
                and "foreach" not in optim_kwargs and params and params[0].is_cuda):
            optim_kwargs["fused"] = True    # one multi-tensor kernel per step on the device
        return OptimCls(params, lr=lr, **optim_kwargs)

    def clip_and_step(self):
        """Gradient-norm clipping + optimizer step (rlpyt/algos/pg/ppo.py:100-104); returns the
        gradient norm (device scalar)."""
        if hasattr(self.optimizer, "clip_and_step"):
            return self.optimizer.clip_and_step(self.clip_grad_norm)
        grad_norm = torch.nn.utils.clip_grad_norm_(self.agent.parameters(), self.clip_grad_norm)
        self.optimizer.step()
        return grad_norm

    # ------------------------------------------------------------------ device helpers
    def on_device(self, x):
        """``x`` on the agent's device (no-op for the HBM-resident sampler's batches)."""
        dev = self.agent.device
        return x if x.device == dev else x.to(dev, non_blocking=True)

    @staticmethod
    def diagnostics_to_host(stats):
        """Rows of per-update scalars (device tensors) -> nested Python lists, one D2H."""
        return torch.stack(stats).cpu().tolist() if stats else []
