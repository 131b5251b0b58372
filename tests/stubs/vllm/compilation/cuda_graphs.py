"""(stand-in) What @support_torch_compile + vLLM's graph backend do for a model at CompilationLevel.PIECEWISE with
full_cuda_graph: a forward over a graph-sized batch runs eagerly `cudagraph_num_of_warmups` times, is captured into a
device graph at the next run of that size (capture_model provokes those runs), and is replayed from then on unless the
forward context says skip_cuda_graphs.  The captured forward INCLUDES attention (no splitting ops in full-graph mode),
so everything attention reads must live in persistent buffers — the runner's job, as in vLLM.

install(model, vllm_config) replaces the instance's `forward`; a later monkeypatch_forward (Ulysses) wraps on top of it,
which is also the order in vLLM (the compiled callable sits inside whatever replaces model.forward)."""
import torch

from vllm.config import CompilationLevel
from vllm.forward_context import get_forward_context

stats = {"captured": 0, "replayed": 0, "eager": 0}
enabled = True      # tests: False = the same padding and control flow, every forward eager (the reference of a replay)


def install(model, vllm_config):
    cc = vllm_config.compilation_config
    inner = model.forward
    state = {}
    pool = [None]

    def forward(*args, **kwargs):
        ids = kwargs.get("input_ids")
        n = ids.shape[0]
        ctx = get_forward_context()
        usable = (enabled and cc.level == CompilationLevel.PIECEWISE and cc.full_cuda_graph and ids.is_cuda
                  and not ctx.skip_cuda_graphs and n in cc.cudagraph_capture_sizes)
        if not usable:
            stats["eager"] += 1
            return inner(*args, **kwargs)
        e = state.setdefault(n, {"warm": 0, "graph": None, "out": None})
        if e["graph"] is not None:
            e["graph"].replay()
            stats["replayed"] += 1
            return e["out"]
        if e["warm"] < cc.cudagraph_num_of_warmups:
            e["warm"] += 1
            stats["eager"] += 1
            return inner(*args, **kwargs)
        g = torch.cuda.CUDAGraph()
        if pool[0] is None:
            pool[0] = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g, pool=pool[0]):
            out = inner(*args, **kwargs)
        e["graph"], e["out"] = g, out
        stats["captured"] += 1
        return out

    model.forward = forward
    model._graph_state = state
    return model
