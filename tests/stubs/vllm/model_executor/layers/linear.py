"""Tensor-parallel linear layers of the stand-in: vLLM's constructor signatures and `(output, bias)` return convention,
weights sharded over get_tp_group() at construction time (so a layer built under the plugin's shift-parallel switch is
sharded over the SP x TP group), `weight_loader`s that take the full checkpoint tensor and keep this rank's slice."""
import torch

from vllm.distributed.parallel_state import get_tp_group


def _param(rows: int, cols: int, dtype=None):
    from vllm.config import get_current_vllm_config
    cfg = get_current_vllm_config()
    dt = dtype or (cfg.model_config.dtype if cfg is not None and cfg.model_config.dtype is not None else torch.float32)
    dev = cfg.device_config.device if cfg is not None else "cpu"
    return torch.nn.Parameter(torch.zeros(rows, cols, dtype=dt, device=dev), requires_grad=False)


class ColumnParallelLinear(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype=None, quant_config=None, output_sizes=None, prefix: str = "",
                 return_bias: bool = True):
        super().__init__()
        assert not bias and not gather_output
        tp = get_tp_group()
        self.tp_size, self.tp_rank = tp.world_size, tp.rank_in_group
        assert output_size % self.tp_size == 0
        self.output_size_per_partition = output_size // self.tp_size
        self.weight = _param(self.output_size_per_partition, input_size, params_dtype)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param, loaded_weight, loaded_shard_id=None):
        n = self.output_size_per_partition
        param.data.copy_(loaded_weight[self.tp_rank * n:(self.tp_rank + 1) * n].to(param.dtype))

    def forward(self, x):
        return x @ self.weight.T, None


class MergedColumnParallelLinear(ColumnParallelLinear):
    def __init__(self, input_size: int, output_sizes, bias: bool = False, gather_output: bool = False, quant_config=None,
                 prefix: str = "", **kw):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias=bias, gather_output=gather_output, prefix=prefix)

    def weight_loader(self, param, loaded_weight, loaded_shard_id=None):
        assert loaded_shard_id is not None
        per = [s // self.tp_size for s in self.output_sizes]
        off = sum(per[:loaded_shard_id])
        n = per[loaded_shard_id]
        param.data[off:off + n].copy_(loaded_weight[self.tp_rank * n:(self.tp_rank + 1) * n].to(param.dtype))


class QKVParallelLinear(torch.nn.Module):
    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int, total_num_kv_heads: int = None,
                 bias: bool = False, skip_bias_add: bool = False, params_dtype=None, quant_config=None, prefix: str = "",
                 return_bias: bool = True):
        super().__init__()
        assert not bias
        tp = get_tp_group()
        self.tp_size, self.tp_rank = tp.world_size, tp.rank_in_group
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_heads if total_num_kv_heads is None else total_num_kv_heads
        self.num_heads = self.total_num_heads // self.tp_size
        if self.tp_size >= self.total_num_kv_heads:
            self.num_kv_heads, self.num_kv_head_replicas = 1, self.tp_size // self.total_num_kv_heads
        else:
            self.num_kv_heads, self.num_kv_head_replicas = self.total_num_kv_heads // self.tp_size, 1
        rows = (self.num_heads + 2 * self.num_kv_heads) * head_size
        self.weight = _param(rows, hidden_size, params_dtype)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param, loaded_weight, loaded_shard_id=None):
        D = self.head_size
        if loaded_shard_id == "q":
            off, n, rank = 0, self.num_heads * D, self.tp_rank
        else:
            n, rank = self.num_kv_heads * D, self.tp_rank // self.num_kv_head_replicas
            off = self.num_heads * D + (n if loaded_shard_id == "v" else 0)
        param.data[off:off + n].copy_(loaded_weight[rank * n:(rank + 1) * n].to(param.dtype))

    def forward(self, x):
        return x @ self.weight.T, None


class RowParallelLinear(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, input_is_parallel: bool = True,
                 skip_bias_add: bool = False, params_dtype=None, reduce_results: bool = True, quant_config=None,
                 prefix: str = "", return_bias: bool = True):
        super().__init__()
        assert not bias and input_is_parallel
        self.tp = get_tp_group()
        self.tp_size, self.tp_rank = self.tp.world_size, self.tp.rank_in_group
        assert input_size % self.tp_size == 0
        self.input_size_per_partition = input_size // self.tp_size
        self.weight = _param(output_size, self.input_size_per_partition, params_dtype)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param, loaded_weight, loaded_shard_id=None):
        n = self.input_size_per_partition
        param.data.copy_(loaded_weight[:, self.tp_rank * n:(self.tp_rank + 1) * n].to(param.dtype))

    def forward(self, x):
        y = x @ self.weight.T
        if self.tp_size > 1:
            y = self.tp.all_reduce(y.float()).to(x.dtype)
        return y, None
