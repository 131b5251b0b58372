"""CPU: the vLLM plugin layer (arcticinference_amd/vllm_plugin/*) executed against the stand-in for vLLM 0.9.2
(tests/stubs/README.md): plugin load, every argument / config / stats / executor / parallel-state patch, and the
patched GPUModelRunner.execute_model driven by a minimal scheduler.  The HIP routes need a GPU
(tests/test_vllm_plugin_gpu.py); here attention and acceptance stay on the stand-in's torch implementations, which is
exactly the path the plugin takes for tensors that are not on the device."""
import os
import numpy as np
import pytest
import torch

import vllm_harness as H

# every vLLM symbol SURVEY.md §8(b) lists as patched (names must keep existing on the target after apply_patch)
PATCHED = {
    "vllm.v1.engine.core:EngineCoreProc": ["run_engine_core"],
    "vllm.v1.worker.worker_base:WorkerBase": ["__init__"],
    "vllm.engine.arg_utils:EngineArgs": ["__new__", "__post_init__", "add_cli_args", "from_cli_args", "create_engine_config",
                                         "_is_v1_supported_oracle"],
    "vllm.engine.arg_utils:AsyncEngineArgs": ["__new__"],
    "vllm.config:ParallelConfig": ["__new__"],
    "vllm.config:SpeculativeConfig": ["__new__", "__post_init__", "from_dict"],
    "vllm.config:VllmConfig": ["__str__"],
    "vllm.config:CompilationConfig": ["set_splitting_ops_for_v1"],      # this build's addition (vllm_plugin/custom_ops.py)
    "vllm.transformers_utils.configs.mlp_speculator:MLPSpeculatorConfig": ["__init__"],
    "vllm.v1.spec_decode.metrics:SpecDecodingStats": ["observe_draft"],
    "vllm.v1.spec_decode.metrics:SpecDecodingLogging": ["log"],
    "vllm.config:ModelConfig": ["get_num_kv_heads", "get_num_attention_heads", "get_layers_start_end_indices"],
    "vllm.distributed.parallel_state": ["initialize_model_parallel", "graph_capture", "_SP", "_SP_TP", "_SP_AA", "_SP_AG"],
    "vllm.v1.executor.multiproc_executor:WorkerProc": ["destroy_model_parallel", "shutdown"],
    "vllm.v1.executor.multiproc_executor:MultiprocExecutor": ["_init_executor"],
    "vllm.attention.layer:Attention": ["__init__", "forward"],
    "vllm.compilation.backends:PiecewiseCompileInterpreter": ["find_symbolic_shape", "call_module"],
    "vllm.model_executor.layers.fused_moe:FusedMoE": ["forward"],
    "vllm.v1.worker.gpu_model_runner:GPUModelRunner": [
        "__init__", "profile_run", "_prepare_inputs", "monkeypatch_forward", "execute_model", "propose_draft_token_ids",
        "propose_arctic_draft_token_ids", "_update_suffix_cache", "propose_suffix_draft_token_ids", "load_model",
        "capture_model", "initialize_kv_cache"],
}


def _resolve(path):
    import importlib
    mod, _, name = path.partition(":")
    m = importlib.import_module(mod)
    return getattr(m, name) if name else m


def test_plugin_load_patches_every_listed_symbol(stub_vllm):
    import vllm.plugins
    from vllm import ModelRegistry
    H.load_plugin(worker=True)
    for path, names in PATCHED.items():
        target = _resolve(path)
        owners = vars(target)["_arctic_patches"]
        for n in names:
            assert n in owners and hasattr(target, n), (path, n)
    assert set(ModelRegistry.models) >= {"ArcticMLPSpeculatorPreTrainedModel", "ArcticLSTMSpeculatorPreTrainedModel",
                                         "MLPVariantSpeculatorPreTrainedModel"}
    # the EngineCore process loads the plugins again before it starts (plugins.py:41-47)
    from vllm.v1.engine.core import EngineCoreProc
    before = vllm.plugins.load_calls
    assert EngineCoreProc.run_engine_core(1, x=2)[0] == "engine core" and vllm.plugins.load_calls == before + 1
    # a second worker in the same process does not re-apply the runner patch; a second plugin load is vLLM's to prevent
    from vllm.config import VllmConfig
    from vllm.v1.worker.worker_base import WorkerBase
    WorkerBase(VllmConfig())
    from arctic_inference.vllm.plugins import arctic_inference_plugin
    with pytest.raises(ValueError):
        arctic_inference_plugin()


def test_plugin_gates(stub_vllm, monkeypatch, caplog):
    from vllm.engine.arg_utils import EngineArgs
    from vllm.platforms import current_platform
    from arctic_inference.vllm.plugins import arctic_inference_plugin
    monkeypatch.setattr(stub_vllm, "__version__", "0.8.5")
    arctic_inference_plugin()
    assert "_arctic_patches" not in vars(EngineArgs)
    monkeypatch.setattr(stub_vllm, "__version__", "0.9.2")
    monkeypatch.setattr(type(current_platform), "rocm", False)
    arctic_inference_plugin()
    assert "_arctic_patches" not in vars(EngineArgs)
    monkeypatch.setattr(type(current_platform), "rocm", True)
    monkeypatch.setenv("VLLM_USE_V1", "0")
    arctic_inference_plugin()
    assert "_arctic_patches" not in vars(EngineArgs)
    monkeypatch.delenv("VLLM_USE_V1")
    arctic_inference_plugin()
    assert "_arctic_patches" in vars(EngineArgs)


def test_engine_args_and_config_patches(stub_vllm):
    from vllm.config import ParallelConfig, SpeculativeConfig
    from vllm.engine.arg_utils import AsyncEngineArgs, EngineArgs
    from vllm.utils import FlexibleArgumentParser
    H.load_plugin(worker=False)
    parser = EngineArgs.add_cli_args(FlexibleArgumentParser())
    ns = parser.parse_args(["--ulysses-sequence-parallel-size", "4", "--enable-shift-parallel", "--shift-parallel-threshold",
                            "256", "--tensor-parallel-size", "2"])
    assert (ns.ulysses_sequence_parallel_size, ns.enable_shift_parallel, ns.shift_parallel_threshold) == (4, True, 256)
    dflt = parser.parse_args([])
    assert (dflt.ulysses_sequence_parallel_size, dflt.enable_shift_parallel, dflt.shift_parallel_threshold) == (1, False, 512)
    ea = EngineArgs.from_cli_args(ns)
    assert type(ea).__name__ == "ArcticEngineArgs" and isinstance(ea, EngineArgs) and ea.post_init_ran
    assert ea.distributed_executor_backend == "mp"          # SP > 1 forces the multiprocess executor (args.py:63-71)
    cfg = ea.create_engine_config()
    pc = cfg.parallel_config
    assert type(pc).__name__ == "ArcticParallelConfig" and pc.world_size == 8 and pc.tensor_parallel_size == 2
    assert (pc.ulysses_sequence_parallel_size, pc.enable_shift_parallel, pc.shift_parallel_threshold) == (4, True, 256)
    assert "ulysses_sequence_parallel_size=4, enable_shift_parallel=True, shift_parallel_threshold=256" in str(cfg)
    assert type(AsyncEngineArgs.from_cli_args(ns)).__name__ == "ArcticAsyncEngineArgs"
    assert type(EngineArgs()).__name__ == "ArcticEngineArgs" and type(AsyncEngineArgs()).__name__ == "ArcticAsyncEngineArgs"
    assert EngineArgs().distributed_executor_backend is None
    # vLLM's V1 oracle does not know "arctic" / "suffix": the patch hides them for the check and puts them back
    ea2 = EngineArgs(speculative_config={"method": "arctic", "num_speculative_tokens": 3})
    assert ea2._is_v1_supported_oracle() is True and ea2.speculative_config["method"] == "arctic"
    assert EngineArgs(speculative_config={"method": "other"})._is_v1_supported_oracle() is False
    # ParallelConfig(...) / SpeculativeConfig(...) construct the Arctic subclasses
    p = ParallelConfig(tensor_parallel_size=2)
    assert type(p).__name__ == "ArcticParallelConfig" and p.world_size == 2 and p.ulysses_sequence_parallel_size == 1
    p.world_size = 99
    assert p.world_size == 2                                 # assignments by vLLM's own code are ignored (config.py:46-52)
    with pytest.raises(ValueError, match="ulysses_sequence_parallel_size must be > 1"):
        type(p)(enable_shift_parallel=True)
    s = SpeculativeConfig(method=None, enable_suffix_decoding=True)
    assert (s.method, s.num_speculative_tokens, s.disable_by_batch_size, s.enable_suffix_decoding) == ("suffix", 64, 64, True)
    s = SpeculativeConfig.from_dict({"method": "arctic", "num_speculative_tokens": 3, "enable_suffix_decoding": True,
                                     "suffix_cache_max_depth": 32})
    assert type(s).__name__ == "ArcticSpeculativeConfig" and s.disable_by_batch_size == 64 and s.suffix_cache_max_depth == 32
    assert SpeculativeConfig(method="ngram", num_speculative_tokens=2).disable_by_batch_size is None
    from vllm.transformers_utils.configs.mlp_speculator import MLPSpeculatorConfig
    assert MLPSpeculatorConfig(vocab_size=10, base_model_arch="LlamaForCausalLM").base_model_arch == "LlamaForCausalLM"


def test_stats_patches(stub_vllm):
    from vllm.v1.spec_decode.metrics import SpecDecodingLogging, SpecDecodingStats
    H.load_plugin(worker=False)
    st = SpecDecodingStats(3)
    st.observe_draft(3, 2)
    st.observe_draft(17, 9)            # a suffix draft, longer than k: unpatched this asserts
    assert st.num_spec_tokens == 17 and len(st.num_accepted_tokens_per_pos) == 17 and st.num_accepted_tokens_per_pos[:3] == [2, 2, 1]
    lg = SpecDecodingLogging()
    lg.log()                           # nothing observed: returns before vLLM's code
    assert lg.logged == 0
    lg.observe(SpecDecodingStats(3))
    lg.observe(st)
    lg.log()                           # ragged per-position lists are padded first
    assert lg.logged == 1 and all(len(x) == 17 for x in lg.accepted_tokens_per_pos_lists)


def test_model_config_executor_and_misc_patches(stub_vllm):
    from vllm.config import ModelConfig, ParallelConfig, VllmConfig
    from vllm.distributed import parallel_state
    from vllm.model_executor.layers.fused_moe import FusedMoE
    from vllm.v1.executor.multiproc_executor import MultiprocExecutor, WorkerProc
    H.load_plugin(worker=False)
    mc = ModelConfig()
    pc = ParallelConfig(tensor_parallel_size=2, ulysses_sequence_parallel_size=2, pipeline_parallel_size=2)
    assert mc.get_num_attention_heads(pc) == 2 and mc.get_num_kv_heads(pc) == 1     # 8 / (2*2), 4 / (2*2)
    pc8 = ParallelConfig(ulysses_sequence_parallel_size=8)
    assert mc.get_num_kv_heads(pc8) == 1 and mc.get_num_attention_heads(pc8) == 1     # never below one head
    pc.rank = 5                                                                       # (pp=1, sp=0, tp=1) of PP x SP x TP
    assert mc.get_layers_start_end_indices(pc) == (1, 2)
    pc.rank = 3
    assert mc.get_layers_start_end_indices(pc) == (0, 1)
    ex = MultiprocExecutor(VllmConfig(parallel_config=pc))
    assert ex.world_size == 8 and len(ex.workers) == 8 and ex.monitor_started and ex.rpc_broadcast_mq.ready
    assert [m[1] for m in WorkerProc.made] == list(range(8)) and WorkerProc.made[0][2].startswith("tcp://127.0.0.1:")
    assert ex.io_thread_pool is not None and ex.output_rank == 6
    ex.io_thread_pool.shutdown()
    bad = ParallelConfig(tensor_parallel_size=2, ulysses_sequence_parallel_size=2)
    object.__setattr__(bad, "pipeline_parallel_size", 1)
    w = WorkerProc()
    parallel_state._SP = parallel_state._SP_TP = None
    w.shutdown()
    assert w.rpc_broadcast_mq is None and parallel_state._SP is None
    assert FusedMoE().forward(torch.ones(2), None).tolist() == [2.0, 2.0]             # forward_impl directly, no custom op


def test_piecewise_interpreter_patch(stub_vllm):
    """A subgraph whose shape symbol is passed at two argument positions (N and a tensor of N rows) compiles with both
    positions recorded; two different symbols are refused."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.fx.experimental.symbolic_shapes import ShapeEnv
    from vllm.compilation.backends import PiecewiseCompileInterpreter
    from vllm.compilation.counter import compilation_counter
    from vllm.config import VllmConfig
    H.load_plugin(worker=False)

    class Sub(torch.nn.Module):
        def forward(self, n, x, m):
            return x * 2

    class Top(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.submod_0 = Sub()

        def forward(self, n, x, m):
            return self.submod_0(n, x, m)

    class LeafTracer(torch.fx.Tracer):          # vLLM's split graph calls its pieces as submodules
        def is_leaf_module(self, m, qualname):
            return isinstance(m, Sub)

    top = Top()
    gm = torch.fx.GraphModule(top, LeafTracer().trace(top))

    class Mgr:
        def compile(self, submod, args, inductor_config, compilation_config, graph_index, num_graphs, runtime_shape):
            self.seen = (graph_index, num_graphs, runtime_shape)
            return submod

    class Backend:
        compiler_manager = Mgr()

    from torch.fx.experimental.symbolic_shapes import DimDynamic, StatelessSymbolicContext
    mode = FakeTensorMode(shape_env=ShapeEnv())
    fake = mode.from_tensor(torch.empty(7, 4), symbolic_context=StatelessSymbolicContext(
        dynamic_sizes=[DimDynamic.DYNAMIC, DimDynamic.STATIC]))
    s = fake.shape[0]                       # the SymInt vLLM would pass as the token count
    assert isinstance(s, torch.SymInt)
    interp = PiecewiseCompileInterpreter(gm, ["submod_0"], VllmConfig(), None, Backend())
    before = compilation_counter.num_piecewise_capturable_graphs_seen
    interp.run(s, fake, s)
    pb = gm.__dict__["submod_0"]
    assert pb.sym_shape_indices == [0, 2] and Backend.compiler_manager.seen == (0, 1, None)
    assert compilation_counter.num_piecewise_capturable_graphs_seen == before + 1


def test_plugin_loader_leaves_the_gpu_runtime_alone_until_the_worker():
    """general plugins are loaded in vLLM's API-server / EngineCore parents, which fork the workers — the
    loader may not initialise (or even load) the HIP library; the worker-side patch does (reference plugins.py:54-63)."""
    import subprocess
    import sys
    code = (
        "import sys, os\n"
        f"sys.path[:0] = [{os.path.dirname(os.path.abspath(__file__))!r}, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}]\n"
        "import vllm_harness as H\n"
        "H.install()\n"
        "from arcticinference_amd import _native\n"
        "H.load_plugin(worker=False)\n"
        "assert _native._lib is None, 'the plugin loader opened libarctic_hip.so'\n"
        "from vllm.engine.arg_utils import EngineArgs\n"
        "assert '_arctic_patches' in vars(EngineArgs)\n"
        "from vllm.config import VllmConfig\n"
        "from vllm.v1.worker.worker_base import WorkerBase\n"
        "WorkerBase(VllmConfig())\n"
        "assert _native._lib is not None, 'the worker patch must load the library and probe the device'\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
