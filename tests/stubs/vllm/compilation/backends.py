import torch


class PiecewiseBackend:
    def __init__(self, graph, vllm_config, graph_pool, piecewise_compile_index, total_piecewise_compiles, sym_shape_indices,
                 compiled_graph_for_general_shape, vllm_backend):
        self.graph, self.sym_shape_indices = graph, sym_shape_indices
        self.compiled_graph_for_general_shape = compiled_graph_for_general_shape

    def __call__(self, *args):
        return self.compiled_graph_for_general_shape(*args)


class PiecewiseCompileInterpreter(torch.fx.Interpreter):
    def __init__(self, module, compile_submod_names, vllm_config, graph_pool, vllm_backend):
        super().__init__(module)
        self.compile_submod_names = compile_submod_names
        self.compilation_config = vllm_config.compilation_config
        self.vllm_config, self.graph_pool, self.vllm_backend = vllm_config, graph_pool, vllm_backend

    def call_module(self, target, args, kwargs):
        return super().call_module(target, args, kwargs)
