"""Rotary position embedding of the stand-in (neox style), vLLM's get_rope / forward(positions, q, k) signature."""
import torch


class RotaryEmbedding(torch.nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position: int, base: float):
        super().__init__()
        self.head_size, self.rotary_dim, self.base = head_size, rotary_dim, base

    def _rotate(self, positions, x):
        n = x.shape[0]
        h = x.view(n, -1, self.head_size).float()
        half = self.rotary_dim // 2
        inv = 1.0 / (self.base ** (torch.arange(0, half, dtype=torch.float32, device=x.device) / half))
        ang = positions.to(torch.float32)[:, None] * inv[None, :]
        cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
        a, b = h[..., :half], h[..., half:2 * half]
        out = torch.cat([a * cos - b * sin, b * cos + a * sin, h[..., 2 * half:]], dim=-1)
        return out.reshape(n, -1).to(x.dtype)

    def forward(self, positions, query, key):
        return self._rotate(positions, query), self._rotate(positions, key)


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, is_neox_style: bool = True, rope_scaling=None,
             dtype=None, partial_rotary_factor: float = 1.0):
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
