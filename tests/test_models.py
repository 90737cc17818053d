"""Model parity with the reference's own classes: tests/golden/models.npz
holds, for every hot-path model built by the REFERENCE under a fixed seed, the parameter names,
shapes and per-tensor checksums, the forward outputs on seeded inputs and the gradients of a
fixed scalar.  CPU tests: this repo's classes expose the same state dict and initialise to the
same values (state dicts interchange).  GPU tests: forward/backward through the HIP input /
conv path reproduce the reference's numbers (fp32 tolerance stated per test)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from model_cases import MODEL_CASES, MODEL_SEED, model_inputs, scalarize  # noqa: E402

CASE_IDS = [c[0] for c in MODEL_CASES]


def build(case):
    name, _ref, mine, kwargs, recurrent = case
    mod, cls = mine.rsplit(".", 1)
    Model = getattr(importlib.import_module(mod), cls)
    torch.manual_seed(MODEL_SEED)
    return Model(image_shape=(4, 104, 80), output_size=6, **kwargs)


@pytest.mark.parametrize("case", MODEL_CASES, ids=CASE_IDS)
def test_state_dict_matches_reference(case):
    g = load_golden("models")
    name = case[0]
    sd = build(case).state_dict()
    assert list(sd.keys()) == list(g[f"{name}_names"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g[f"{name}_shapes"])
    # same seed, same construction order => bit-identical initial parameters
    np.testing.assert_array_equal([v.double().sum().item() for v in sd.values()],
                                  g[f"{name}_sums"])
    np.testing.assert_array_equal([v.double().abs().sum().item() for v in sd.values()],
                                  g[f"{name}_abs_sums"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", MODEL_CASES, ids=CASE_IDS)
def test_forward_backward_matches_reference(case):
    """Outputs within 2e-5 (abs, O(1) values; softmax outputs 1e-6) and per-parameter gradient
    norms within 1e-4 relative of the reference's CPU fp32 run: the device path reorders the
    fp32 sums of the convolutions (MFMA / MIOpen) and of the 3456..512-long dot products."""
    g = load_golden("models")
    name, recurrent = case[0], case[4]
    model = build(case).cuda()
    inputs = model_inputs(recurrent)
    dev_inputs = tuple(tuple(y.cuda() for y in x) if isinstance(x, tuple) else x.cuda()
                       for x in inputs)
    res = model(*dev_inputs)
    res = res if isinstance(res, tuple) else (res,)
    scalarize(res).backward()
    k = 0
    for o in res:
        for leaf in ([o] if isinstance(o, torch.Tensor) else list(o)):
            ref = g[f"{name}_out{k}"]
            got = leaf.detach().cpu().numpy().reshape(ref.shape)
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5)
            k += 1
    norms = np.array([p.grad.double().norm().item() for p in model.parameters()])
    # atol: the dueling heads' advantage_bias has a mathematically zero gradient (mean-centred)
    np.testing.assert_allclose(norms, g[f"{name}_grad_norms"], rtol=1e-4, atol=1e-5)
    for n, p in model.named_parameters():
        key = f"{name}_grad__{n}"
        if key in g:
            ref = g[key]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3,
                                       atol=2e-5 * max(np.abs(ref).max(), 0.05))
