"""``ClipAdam``: torch.optim.Adam whose update, together with the gradient-norm clipping that
precedes it in every algorithm of this path, runs as TWO kernel launches for the whole model
(``rlpyt_clip_adam_step_f32``, csrc/optim.hip).

The reference does, per minibatch (rlpyt/algos/pg/ppo.py:100-104, a2c.py:52-56, dqn/dqn.py:176-180)::

    grad_norm = torch.nn.utils.clip_grad_norm_(self.agent.parameters(), self.clip_grad_norm)
    self.optimizer.step()

-- on the device that is a dozen launch-bound kernels for 7 MB of parameters.  The algorithms here
call ``optimizer.clip_and_step(max_norm)`` when the optimizer offers it and fall back to the two
reference statements otherwise (any other ``OptimCls`` keeps working).

This is a subclass of ``torch.optim.Adam``: same constructor, same ``param_groups`` (LR schedulers
work unchanged), same ``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so
optimizer snapshots interchange with the reference's.  ``step()`` alone is torch's own.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import AdamTensor, check, lib

MAX_TENSORS = 32


class ClipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kwargs):
        kwargs.pop("fused", None)       # this IS the fused path
        kwargs.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kwargs)
        self._ws = None
        self._norm = None

    def supports_fused(self):
        if len(self.param_groups) != 1 or len(self.param_groups[0]["params"]) > MAX_TENSORS:
            return False
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or group.get("decoupled_weight_decay"):
                return False
            for p in group["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    return False
        return True

    @torch.no_grad()
    def launch_captured_step(self, max_norm, hyper_dev, tick_ctr):
        """The two launches of ``clip_and_step`` with the update's ``lr / bc1`` and ``1 / sqrt(bc2)``
        read from ``hyper_dev`` (device, 2 floats) at run time and ``tick_ctr`` (device int64)
        advanced at the end -- for use INSIDE a captured update graph.  No host-side step counting:
        the caller adds the number of replays to every ``state[p]["step"]`` (``advance_steps``).
        Requires ``supports_fused()``, every parameter with a gradient and initialised state."""
        group = self.param_groups[0]
        rows = []
        for p in group["params"]:
            st = self.state[p]
            g = p.grad
            assert g is not None and len(st) and g.is_contiguous() and g.dtype == torch.float32
            rows.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
        assert 0 < len(rows) <= MAX_TENSORS
        table = (AdamTensor * len(rows))()
        for k, (p, g, m, v) in enumerate(rows):
            table[k] = AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
        b1, b2 = group["betas"]
        check(lib.rlpyt_clip_adam_step_dev_f32(
            ctypes.cast(table, ctypes.c_void_p), len(rows), float(group["lr"]), float(b1), float(b2),
            float(group["eps"]), float(group["weight_decay"]), 1,
            float(max_norm) if max_norm else 0., ctypes.c_void_p(self._ws.data_ptr()),
            ctypes.c_void_p(self._norm.data_ptr()), ctypes.c_void_p(hyper_dev.data_ptr()),
            ctypes.c_void_p(tick_ctr.data_ptr()),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "rlpyt_clip_adam_step_dev_f32")
        self._keep = rows
        return self._norm

    @staticmethod
    def _to_out(norm, norm_out):
        if norm_out is None:
            return norm
        norm_out.copy_(norm.reshape(norm_out.shape))
        return norm_out

    def captured_ready(self):
        """True once ``launch_captured_step`` may be captured: fused path, every parameter has state
        (an eager update ran) on one common step count."""
        if not self.supports_fused() or self._ws is None:
            return False
        steps = set()
        for p in self.param_groups[0]["params"]:
            st = self.state.get(p, {})
            if not len(st) or st["step"].is_cuda:
                return False
            steps.add(int(st["step"].item()))
        return len(steps) == 1

    def common_step(self):
        p = self.param_groups[0]["params"][0]
        return int(self.state[p]["step"].item())

    def hyper_rows(self, n_updates):
        """[(lr / bc1, 1 / sqrt(bc2))] of the next ``n_updates`` updates -- the eager path's
        double-precision host arithmetic (csrc/optim.hip), one row per update."""
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        lr, step0 = float(group["lr"]), self.common_step()
        rows = []
        for k in range(1, n_updates + 1):
            bc1, bc2 = 1.0 - b1 ** (step0 + k), 1.0 - b2 ** (step0 + k)
            rows.append((lr / bc1, 1.0 / math.sqrt(bc2)))
        return rows

    def advance_steps(self, n_updates):
        """Host-side bookkeeping of ``n_updates`` captured updates."""
        params = self.param_groups[0]["params"]
        for p in params:
            self.state[p]["step"] += n_updates
        torch.autograd.graph.increment_version(list(params))
        self._opt_called = True

    @torch.no_grad()
    def clip_and_step(self, max_norm, norm_out=None):
        """``clip_grad_norm_(params, max_norm)`` + ``step()``; returns the total gradient norm
        (device scalar tensor, before clipping) like ``clip_grad_norm_``.  ``max_norm`` None or
        <= 0: no clipping.  ``norm_out``: a one-element f32 device tensor (e.g. a slot of the caller's
        diagnostics table) that receives the norm instead of the optimizer's own scalar."""
        if not self.supports_fused():
            params = [p for g in self.param_groups for p in g["params"]]
            norm = (torch.nn.utils.clip_grad_norm_(params, max_norm) if max_norm
                    else torch.zeros((), device=params[0].device))
            self.step()
            return self._to_out(norm, norm_out)
        _lib.require_gpu()
        dev = self.param_groups[0]["params"][0].device
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(lib.rlpyt_clip_adam_workspace_bytes(), dtype=torch.uint8,
                                   device=dev)
            self._norm = torch.zeros((), dtype=torch.float32, device=dev)
        # the clip norm is over ALL parameters (clip_grad_norm_ semantics); a call updates up to
        # MAX_TENSORS tensors of one hyper-parameter group
        group = self.param_groups[0]
        rows = []
        steps = set()
        for p in group["params"]:
            if p.grad is None:
                continue
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif st["step"].is_cuda:
                # a snapshot written by a fused=True Adam holds `step` on the device: bring it
                # to the host ONCE (reading it there every update would sync the stream)
                st["step"] = st["step"].detach().to("cpu", torch.float32)
            steps.add(int(st["step"].item()))
        if len(steps) > 1:
            # parameters with different step counts (a gradient that was None in earlier updates,
            # a merged state dict) need per-tensor bias corrections: the kernel applies ONE, so
            # this update goes through torch's own Adam
            params = [p for p in group["params"] if p.grad is not None]
            norm = (torch.nn.utils.clip_grad_norm_(params, max_norm) if max_norm
                    else torch.zeros((), device=dev))
            self.step()
            return self._to_out(norm, norm_out)
        step = (steps.pop() + 1) if steps else None
        for p in group["params"]:
            if p.grad is None:
                continue
            g = p.grad
            if not (g.is_contiguous() and g.dtype == torch.float32):
                g = g.contiguous().float()
            st = self.state[p]
            st["step"] += 1
            rows.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
        if not rows:
            return self._to_out(self._norm.zero_(), norm_out)
        assert len(rows) <= MAX_TENSORS, f"ClipAdam: {len(rows)} tensors > {MAX_TENSORS}"
        table = (AdamTensor * len(rows))()
        # a parameter whose transpose somebody keeps (ops.TransposedMirror: the update trunk's weight,
        # read transposed by its input-gradient GEMM) gets its new W^T from this launch
        from .ops import TransposedMirror
        mirror_k, mirror_t = -1, None
        for k, (p, g, m, v) in enumerate(rows):
            table[k] = AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
            if (mirror_t is None and p.dim() == 2 and p.shape[0] % 32 == 0 and p.shape[1] % 32 == 0
                    and ((p.data_ptr() | g.data_ptr() | m.data_ptr() | v.data_ptr()) & 15) == 0):
                mt = TransposedMirror.buffer_for(p)
                if mt is not None:
                    mirror_k, mirror_t = k, mt
        b1, b2 = group["betas"]
        lr = group["lr"]
        norm_dst = self._norm if norm_out is None else norm_out
        check(lib.rlpyt_clip_adam_step_mirror_f32(
            ctypes.cast(table, ctypes.c_void_p), len(rows), float(lr), float(b1), float(b2),
            float(group["eps"]), float(group["weight_decay"]), step,
            float(max_norm) if max_norm else 0., ctypes.c_void_p(self._ws.data_ptr()),
            ctypes.c_void_p(norm_dst.data_ptr()), None, None, mirror_k,
            None if mirror_t is None else ctypes.c_void_p(mirror_t.data_ptr()),
            rows[mirror_k][0].shape[0] if mirror_t is not None else 0,
            rows[mirror_k][0].shape[1] if mirror_t is not None else 0,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "rlpyt_clip_adam_step_mirror_f32")
        self._keep = rows          # the launch is asynchronous: keep the gradient tensors alive
        # the kernel wrote the parameters through raw pointers: tell autograd / every cache keyed
        # on ``Tensor._version`` (ops.LstmStep's concatenated weight buffer) that they changed
        torch.autograd.graph.increment_version([r[0] for r in rows])
        if mirror_t is not None:
            TransposedMirror.written(rows[mirror_k][0])
        if norm_out is not None:
            self._opt_called = True
            return norm_out
        self._opt_called = True    # (torch's LR schedulers check that a step preceded theirs)
        return self._norm
