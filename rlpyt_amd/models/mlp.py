"""MLP with the module/parameter naming of rlpyt/models/mlp.py:5-46 (``model.<i>``)."""
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

    def forward(self, input):
        return self.model(input)

    @property
    def output_size(self):
        return self._output_size
