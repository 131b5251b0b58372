"""Config dataclasses of the stand-in: the fields the plugin reads or patches."""
import dataclasses
from dataclasses import dataclass, field
from typing import Any, Optional


def config(cls):
    return cls


class CompilationLevel:
    NO_COMPILATION = 0
    PIECEWISE = 3


@dataclass
class PassConfig:
    enable_sequence_parallelism: bool = False


@dataclass
class CompilationConfig:
    level: int = 0
    pass_config: PassConfig = field(default_factory=PassConfig)
    cudagraph_num_of_warmups: int = 1
    cudagraph_capture_sizes: tuple = (8, 4, 2, 1)
    inductor_compile_config: dict = field(default_factory=dict)
    static_forward_context: dict = field(default_factory=dict)
    full_cuda_graph: bool = False
    splitting_ops: list = field(default_factory=list)

    def set_splitting_ops_for_v1(self):
        """The ops vLLM cuts its piecewise graphs at (attention runs eagerly between the captured pieces)."""
        if self.splitting_ops and self.full_cuda_graph:
            raise ValueError("full_cuda_graph cannot be used together with splitting_ops")
        if not self.splitting_ops:
            self.splitting_ops = [] if self.full_cuda_graph else ["vllm.unified_attention",
                                                                   "vllm.unified_attention_with_output"]


@dataclass
class ParallelConfig:
    pipeline_parallel_size: int = 1
    tensor_parallel_size: int = 1
    data_parallel_size: int = 1
    distributed_executor_backend: Optional[str] = None
    rank: int = 0
    sd_worker_cls: str = "auto"
    worker_cls: str = "auto"
    world_size: int = dataclasses.field(init=False, default=1)

    def __post_init__(self):
        self.world_size = self.pipeline_parallel_size * self.tensor_parallel_size
        if self.distributed_executor_backend is None and self.world_size > 1:
            self.distributed_executor_backend = "mp"


# Python >= 3.10: a classmethod object exposes the plain function as `__wrapped__`, which is what the plugin unwraps
# (`SpeculativeConfig.__dict__["from_dict"].__wrapped__`)
_classmethod_with_wrapped = classmethod


@dataclass
class SpeculativeConfig:
    method: Optional[str] = None
    num_speculative_tokens: Optional[int] = None
    model: Optional[str] = None
    disable_by_batch_size: Optional[int] = None
    draft_model_config: Any = None
    draft_parallel_config: Any = None
    verified: int = dataclasses.field(init=False, default=0)

    def __post_init__(self):
        self._verify_args()

    def _verify_args(self):
        if self.num_speculative_tokens is not None and self.num_speculative_tokens <= 0:
            raise ValueError("num_speculative_tokens must be positive")
        self.verified += 1

    @_classmethod_with_wrapped
    def from_dict(cls, dict_value: dict):
        return cls(**dict_value)


@dataclass
class HfConfig:
    model_type: str = "toy_llama"
    architectures: list = field(default_factory=lambda: ["ToyLlamaForCausalLM"])
    num_hidden_layers: int = 2
    num_attention_heads: int = 8
    num_key_value_heads: int = 4
    hidden_size: int = 512
    head_dim: int = 64
    vocab_size: int = 2000
    seed: int = 0
    sliding_window: Any = None        # gpt-oss-like: every other layer attends over the last `sliding_window` tokens
    attention_sinks: bool = False     # ... and every layer has one learned sink logit per head


@dataclass
class ModelConfig:
    hf_config: Any = field(default_factory=HfConfig)
    max_model_len: int = 512
    dtype: Any = None
    is_multimodal_model: bool = False
    uses_mrope: bool = False

    @property
    def hf_text_config(self):
        return self.hf_config

    @property
    def architectures(self):
        return self.hf_config.architectures

    def get_hidden_size(self) -> int:
        return self.hf_config.hidden_size

    def get_vocab_size(self) -> int:
        return self.hf_config.vocab_size

    def get_head_size(self) -> int:
        return self.hf_config.head_dim

    def get_num_kv_heads(self, parallel_config) -> int:
        return max(1, self.hf_config.num_key_value_heads // parallel_config.tensor_parallel_size)

    def get_num_attention_heads(self, parallel_config) -> int:
        return self.hf_config.num_attention_heads // parallel_config.tensor_parallel_size

    def get_layers_start_end_indices(self, parallel_config):
        from vllm.distributed.utils import get_pp_indices
        pp_rank = (parallel_config.rank // parallel_config.tensor_parallel_size) % parallel_config.pipeline_parallel_size
        return get_pp_indices(self.hf_config.num_hidden_layers, pp_rank, parallel_config.pipeline_parallel_size)


@dataclass
class SchedulerConfig:
    max_num_seqs: int = 8
    max_num_batched_tokens: int = 1024


@dataclass
class CacheConfig:
    block_size: int = 16
    cache_dtype: str = "auto"
    num_gpu_blocks: int = 256


@dataclass
class DeviceConfig:
    device: Any = "cpu"


@dataclass
class LoadConfig:
    load_format: str = "dummy"


@dataclass
class VllmConfig:
    model_config: ModelConfig = field(default_factory=ModelConfig)
    parallel_config: Any = field(default_factory=ParallelConfig)
    scheduler_config: SchedulerConfig = field(default_factory=SchedulerConfig)
    cache_config: CacheConfig = field(default_factory=CacheConfig)
    speculative_config: Any = None
    compilation_config: CompilationConfig = field(default_factory=CompilationConfig)
    device_config: DeviceConfig = field(default_factory=DeviceConfig)
    load_config: LoadConfig = field(default_factory=LoadConfig)
    quant_config: Any = None

    def __post_init__(self):
        self.compilation_config.set_splitting_ops_for_v1()

    def __str__(self):
        return f"model={self.model_config.hf_config.model_type}, tensor_parallel_size={self.parallel_config.tensor_parallel_size}"

    def pad_for_cudagraph(self, batch_size: int) -> int:
        for s in sorted(self.compilation_config.cudagraph_capture_sizes):
            if s >= batch_size:
                return s
        return batch_size

    @staticmethod
    def _get_quantization_config(model_config, load_config):
        return None


_current: Optional[VllmConfig] = None


def get_current_vllm_config() -> Optional[VllmConfig]:
    return _current


def set_current_vllm_config(cfg: Optional[VllmConfig]) -> None:
    global _current
    _current = cfg
