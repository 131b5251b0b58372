"""EngineArgs / AsyncEngineArgs patches: the three Arctic flags on vLLM's command line and engine-argument
dataclasses (/root/reference/arctic_inference/vllm/args.py:29-148).

    --ulysses-sequence-parallel-size N   --enable-shift-parallel   --shift-parallel-threshold T

They reach ParallelConfig in create_engine_config, which rebuilds the parallel config as an ArcticParallelConfig
(vLLM's own code constructs it before it knows the extra fields)."""
from __future__ import annotations

import dataclasses

from .config import ArcticArgs

_built = None


def build_args_patches():
    global _built
    if _built is not None:
        return _built
    from vllm.engine.arg_utils import AsyncEngineArgs, EngineArgs

    from ..patching import ArcticPatch
    from .config import arctic_parallel_config_class

    @dataclasses.dataclass
    class ArcticEngineArgs(EngineArgs, ArcticArgs):
        pass

    @dataclasses.dataclass
    class ArcticAsyncEngineArgs(AsyncEngineArgs, ArcticArgs):
        pass

    def _default_executor(self) -> None:
        # ParallelConfig never sees the Ulysses size in its constructor and would pick the single-process executor
        if self.ulysses_sequence_parallel_size > 1 and self.distributed_executor_backend is None:
            self.distributed_executor_backend = "mp"

    class EngineArgsPatch(ArcticPatch[EngineArgs]):
        _orig_post_init = EngineArgs.__post_init__
        _orig_add_cli_args = EngineArgs.add_cli_args
        _orig_from_cli_args = EngineArgs.__dict__["from_cli_args"].__wrapped__
        _orig_create_engine_config = EngineArgs.create_engine_config
        _orig_is_v1_supported_oracle = EngineArgs._is_v1_supported_oracle

        def __new__(cls, *args, **kwargs):
            if cls is EngineArgs:            # EngineArgs(...) makes the Arctic subclass
                return ArcticEngineArgs.__new__(ArcticEngineArgs, *args, **kwargs)
            return super(EngineArgs, cls).__new__(cls)

        def __post_init__(self):
            _default_executor(self)
            self._orig_post_init()

        @staticmethod
        def add_cli_args(parser):
            parser = EngineArgsPatch._orig_add_cli_args(parser)
            group = parser.add_argument_group(title="Arctic Inference", description="Arctic Inference configuration.")
            group.add_argument("--ulysses-sequence-parallel-size", type=int,
                               default=ArcticArgs.ulysses_sequence_parallel_size,
                               help="Number of Ulysses sequence parallel replicas")
            group.add_argument("--enable-shift-parallel", action="store_true", help="If True, enable shift parallelism.")
            group.add_argument("--shift-parallel-threshold", type=int, default=ArcticArgs.shift_parallel_threshold,
                               help="Ulysses sequence parallel if batch size > threshold, otherwise tensor parallel across "
                                    "the whole world size")
            return parser

        @classmethod
        def from_cli_args(cls, args):
            target = {EngineArgs: ArcticEngineArgs, AsyncEngineArgs: ArcticAsyncEngineArgs}.get(cls, cls)
            return EngineArgsPatch._orig_from_cli_args(target, args)

        def create_engine_config(self, *args, **kwargs):
            _default_executor(self)
            vllm_config = self._orig_create_engine_config(*args, **kwargs)
            pc = vllm_config.parallel_config
            fields = {f.name: getattr(pc, f.name) for f in dataclasses.fields(pc) if f.init}
            fields.update(ulysses_sequence_parallel_size=self.ulysses_sequence_parallel_size,
                          enable_shift_parallel=self.enable_shift_parallel,
                          shift_parallel_threshold=self.shift_parallel_threshold)
            vllm_config.parallel_config = arctic_parallel_config_class()(**fields)
            return vllm_config

        def _is_v1_supported_oracle(self, *args, **kwargs):
            # vLLM's oracle rejects speculative methods it does not know; this plugin is V1-only and was gated on that
            keep = self.speculative_config
            if keep is not None and keep.get("method") in ("arctic", "suffix"):
                self.speculative_config = None
            try:
                return self._orig_is_v1_supported_oracle(*args, **kwargs)
            finally:
                self.speculative_config = keep

    class AsyncEngineArgsPatch(ArcticPatch[AsyncEngineArgs]):
        def __new__(cls, *args, **kwargs):
            if cls is AsyncEngineArgs:
                return ArcticAsyncEngineArgs.__new__(ArcticAsyncEngineArgs, *args, **kwargs)
            return super(AsyncEngineArgs, cls).__new__(cls)

    _built = [EngineArgsPatch, AsyncEngineArgsPatch]
    return _built
