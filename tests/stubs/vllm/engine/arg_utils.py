import argparse
import dataclasses
from dataclasses import dataclass
from typing import Optional

from vllm.config import (CacheConfig, ModelConfig, ParallelConfig, SchedulerConfig, SpeculativeConfig, VllmConfig,
                         _classmethod_with_wrapped)


@dataclass
class EngineArgs:
    model: str = "toy"
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1
    distributed_executor_backend: Optional[str] = None
    max_model_len: int = 512
    speculative_config: Optional[dict] = None

    def __post_init__(self):
        self.post_init_ran = True

    @staticmethod
    def add_cli_args(parser):
        parser.add_argument("--model", type=str, default="toy")
        parser.add_argument("--tensor-parallel-size", type=int, default=1)
        parser.add_argument("--pipeline-parallel-size", type=int, default=1)
        parser.add_argument("--distributed-executor-backend", type=str, default=None)
        parser.add_argument("--max-model-len", type=int, default=512)
        return parser

    @_classmethod_with_wrapped
    def from_cli_args(cls, args: argparse.Namespace):
        names = [f.name for f in dataclasses.fields(cls)]
        return cls(**{n: getattr(args, n) for n in names if hasattr(args, n)})

    def create_engine_config(self, usage_context=None) -> VllmConfig:
        pc = ParallelConfig(pipeline_parallel_size=self.pipeline_parallel_size, tensor_parallel_size=self.tensor_parallel_size,
                            distributed_executor_backend=self.distributed_executor_backend)
        sc = SpeculativeConfig.from_dict(self.speculative_config) if self.speculative_config else None
        return VllmConfig(model_config=ModelConfig(max_model_len=self.max_model_len), parallel_config=pc,
                          scheduler_config=SchedulerConfig(), cache_config=CacheConfig(), speculative_config=sc)

    def _is_v1_supported_oracle(self, model_config=None) -> bool:
        sc = self.speculative_config
        return sc is None or sc.get("method") in ("ngram", "eagle", "medusa")


@dataclass
class AsyncEngineArgs(EngineArgs):
    disable_log_requests: bool = False
