"""RMSNorm of the stand-in: vLLM's call convention (with a residual it returns the pair, the sum becoming the residual)."""
import torch


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        from vllm.model_executor.layers.linear import _param
        self.weight = _param(1, hidden_size)
        self.weight.data = torch.ones_like(self.weight.data[0])
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor, residual: torch.Tensor = None):
        if residual is not None:
            x = x + residual
            residual = x
        xf = x.float()
        y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)).to(x.dtype) * self.weight
        return y if residual is None else (y, residual)
