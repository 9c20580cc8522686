"""Seeded parameter / index generators shared by tests/golden/make_golden.py (build container, with the reference imported) and the tests that
consume tests/golden/reference_golden_heads.npz (no reference anywhere near): the fixture then only has to hold inputs, outputs and gradient
samples -- the (large) parameter tensors are regenerated from the seed on both sides with torch's CPU generator."""
import torch


def fill_params(module, seed):
    """every parameter of `module` from one CPU generator, in named_parameters() order: weights ~ N(0, 1 / fan_in) (the scaling layer x 0.3),
    biases ~ N(0, 0.05^2) -- non-trivial recurrent dynamics without saturating the gates"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            r = torch.randn(p.shape, generator=g, dtype=torch.float32)
            if 'bias' in name:
                p.copy_(r * 0.05)
            else:
                p.copy_(r * (p.shape[-1] ** -0.5) * (0.3 if 'scaling_layer' in name else 1.0))
    return module


def sample_index(numel, k, seed):
    """k distinct positions of a flattened tensor (all of them when numel <= k), ascending"""
    if numel <= k:
        return torch.arange(numel)
    return torch.randperm(numel, generator=torch.Generator().manual_seed(seed))[:k].sort().values


GRAD_SAMPLES = 1024          # entries kept per parameter-gradient tensor
SCORE_SAMPLES = 8192        # entries kept per utterance of the (B, n_params) scoring matrix
