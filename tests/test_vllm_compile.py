"""CPU: the patched Attention.forward under Dynamo.  vLLM V1 traces the model with fullgraph=True at its default
compilation level; the plugin's forward is a ctypes / host-logic route, so it must present itself to the tracer as ONE
custom op with a fake implementation (vllm_plugin/custom_ops.py) and be listed in the ops vLLM cuts its
piecewise graphs at.  Here the stand-in's toy model is compiled with fullgraph=True (any graph break raises) and driven
through the patched execute_model: same tokens as the eager patched run, the op in every captured graph."""
import torch

import vllm_harness as H
from test_vllm_runner import build_runner, drive, prompts_for


def test_patched_attention_traces_as_one_custom_op_and_splits_the_graph(stub_vllm):
    from vllm.config import SpeculativeConfig
    H.load_plugin()
    from arcticinference_amd.vllm_plugin.custom_ops import SPLITTING_OP
    P = prompts_for(7, 3)
    spec = lambda: SpeculativeConfig(method="ngram", num_speculative_tokens=2)
    # the compiled runner goes FIRST: vLLM's first forward (the profile run) is already traced, so nothing the patched
    # forward needs may be created lazily inside it
    r = build_runner(spec=spec())
    assert r.vllm_config.compilation_config.splitting_ops[-1] == SPLITTING_OP
    assert "vllm.unified_attention" in r.vllm_config.compilation_config.splitting_ops
    graphs = []

    def backend(gm, example_inputs):
        graphs.append(gm)
        return gm.forward

    torch._dynamo.reset()
    r.model.forward = torch.compile(r.model.forward, fullgraph=True, backend=backend, dynamic=True)
    got, _ = drive(r, P, 8)
    want, _ = drive(build_runner(spec=spec()), P, 8)
    assert got == want
    assert graphs, "the model was never traced"
    for gm in graphs:
        targets = [n.target for n in gm.graph.nodes if n.op == "call_function"]
        ours = [t for t in targets if "arctic_inference" in str(t) and "attention" in str(t)]
        assert len(ours) == 2, targets          # one op per toy layer, nothing of the route leaked into the graph
    torch._dynamo.reset()


def test_full_graph_mode_keeps_vllms_empty_split_list(stub_vllm):
    from vllm.config import CompilationConfig, VllmConfig
    H.load_plugin()
    cfg = VllmConfig(compilation_config=CompilationConfig(level=3, full_cuda_graph=True))
    assert cfg.compilation_config.splitting_ops == []
