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
        return m(input)

    @property
    def output_size(self):
        return self._output_size
