"""MLP with the module/parameter naming of rlpyt/models/mlp.py:5-46 (``model.<i>``)."""
import os

import torch


class MlpModel(torch.nn.Module):
    def __init__(self, input_size, hidden_sizes, output_size=None, nonlinearity=torch.nn.ReLU):
        super().__init__()
        if isinstance(hidden_sizes, int):
            hidden_sizes = [hidden_sizes]
        hidden_sizes = list(hidden_sizes or [])
        layers, n_in = [], input_size
        for n_out in hidden_sizes:
            layers += [torch.nn.Linear(n_in, n_out), nonlinearity()]
            n_in = n_out
        if output_size is not None:
            layers.append(torch.nn.Linear(n_in, output_size))
        self.model = torch.nn.Sequential(*layers)
        self._output_size = n_in if output_size is None else output_size

    # set False (or RLPYT_Q_HEAD=0) for the library GEMMs in no-grad forwards too (A/B tests)
    use_fused_q_head = os.environ.get("RLPYT_Q_HEAD", "1") != "0"
    # ... and RLPYT_Q_HEAD_TRAIN=0 for autograd through the library GEMMs in the pass under autograd
    use_fused_q_head_train = os.environ.get("RLPYT_Q_HEAD_TRAIN", "1") != "0"

    def forward(self, input):
        m = self.model
        if (self.use_fused_q_head_train and len(m) == 3 and torch.is_grad_enabled()
                and isinstance(input, torch.Tensor) and input.is_cuda and input.dim() == 2
                and isinstance(m[1], torch.nn.ReLU) and isinstance(m[0], torch.nn.Linear)
                and isinstance(m[2], torch.nn.Linear)):
            # the same head in the online network's pass of an update: own forward that keeps the hidden
            # activations, own backward for everything but the hidden layer's two GEMMs
            from .. import ops
            if ops.mlp_q_head_train_ok(input, m[0], m[2]):
                return ops.mlp_q_head_train(input, m[0], m[2])
        if (self.use_fused_q_head and len(m) == 3 and not torch.is_grad_enabled()
                and isinstance(input, torch.Tensor) and input.is_cuda and input.dim() == 2
                and isinstance(m[1], torch.nn.ReLU) and isinstance(m[0], torch.nn.Linear)
                and isinstance(m[2], torch.nn.Linear)):
            # Linear -> ReLU -> Linear with few outputs on a sampling / target batch: the Q-value
            # heads of the DQN family (split-K hidden layer + one finishing kernel)
            from .. import ops
            if ops.mlp_q_head_ok(input, m[0], m[2]):
                return ops.mlp_q_head(input.contiguous(), m[0], m[2])
        if (self.use_split_gemm and torch.is_grad_enabled() and isinstance(input, torch.Tensor)
                and input.is_cuda and input.dim() == 2 and input.dtype == torch.float32
                and input.shape[0] >= self.SPLIT_GEMM_ROWS):
            return self._forward_wide(input)
        return m(input)

    # set False (or RLPYT_MLP_GEMM=0 / RLPYT_SPLIT_GEMM=0) for the library f32 GEMMs at update-size row counts
    # (A/B tests)
    use_split_gemm = (os.environ.get("RLPYT_MLP_GEMM", "1") != "0"
                      and os.environ.get("RLPYT_SPLIT_GEMM", "1") != "0")
    SPLIT_GEMM_ROWS = 1024

    def _forward_wide(self, x):
        """The pass under autograd at update-size row counts (R2D1's 5 440 rows through the trunk FC
        6912 -> 512): every Linear whose sizes ``ops.linear_nobias`` covers takes its input and
        weight gradients on the bf16x6 GEMMs (``gemm_nt`` / ``gemm_tn``, f32-level error: 202 / 199
        against 329 / 314 us for the library's f32 GEMMs) and its forward wherever that kernel fills
        the chip; the other layers as they are."""
        from .. import ops
        for layer in self.model:
            if (isinstance(layer, torch.nn.Linear) and x.shape[0] % 32 == 0
                    and layer.in_features % 32 == 0 and layer.out_features % 32 == 0
                    and layer.in_features >= 256 and layer.out_features >= 256
                    and layer.weight.dtype == torch.float32):
                x = ops.linear_nobias(x.contiguous(), layer.weight)
                if layer.bias is not None:
                    x = x + layer.bias
            else:
                x = layer(x)
        return x

    @property
    def output_size(self):
        return self._output_size
