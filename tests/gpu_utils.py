"""Helpers shared by the `-m gpu` parity tests."""
import torch

BF16 = torch.bfloat16


def to_act(x, dev):
    """NCDHW fp32 (cpu) -> bf16 channels-last activation on the GPU."""
    return x.to(dev).to(BF16).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)


def to_w(w, dev):
    return w.to(dev).to(BF16).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)


def rb(x):
    """bf16 rounding as fp32 (what the kernels actually see)."""
    return x.to(BF16).float()


def rel_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    denom = ref.abs().max().clamp_min(1e-20)
    return float((got - ref).abs().max() / denom)


def rel_l2(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-20))


def assert_close(got, ref, tol, what=""):
    e = rel_err(got, ref)
    assert e <= tol, f"{what}: max-normalised error {e:.3e} > {tol:.1e} (l2 {rel_l2(got, ref):.3e})"
    return e
