"""Flag and config names / defaults of the reference (args.py:29-34, config.py:27-62, :76-102), as plain
dataclasses that need no vLLM, plus the builders of the vLLM patches that graft them onto EngineArgs /
ParallelConfig / SpeculativeConfig when vLLM is importable."""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Optional

logger = logging.getLogger(__name__)


@dataclass
class ArcticArgs:
    """CLI: --ulysses-sequence-parallel-size, --enable-shift-parallel, --shift-parallel-threshold (args.py:29-34,:80-96)."""
    ulysses_sequence_parallel_size: int = 1
    enable_shift_parallel: bool = False
    shift_parallel_threshold: int = 512


@dataclass
class ArcticParallelSettings(ArcticArgs):
    pipeline_parallel_size: int = 1
    tensor_parallel_size: int = 1

    def __post_init__(self):
        if self.enable_shift_parallel and self.ulysses_sequence_parallel_size == 1:
            # config.py:34-39
            raise ValueError("ulysses_sequence_parallel_size must be > 1 when enable_shift_parallel is True.")

    @property
    def world_size(self) -> int:  # config.py:41-44: PP * TP * SP
        return self.pipeline_parallel_size * self.tensor_parallel_size * self.ulysses_sequence_parallel_size

    def distributed_executor_backend(self, requested: Optional[str]) -> Optional[str]:
        # args.py:63-71: SP > 1 forces the multiprocess executor unless one was chosen
        if self.ulysses_sequence_parallel_size > 1 and requested is None:
            return "mp"
        return requested


@dataclass
class ArcticSpeculativeSettings:
    """speculative_config keys (config.py:55-62) and the defaulting rules of SpeculativeConfigPatch.__post_init__ (:89-106)."""
    method: Optional[str] = None
    num_speculative_tokens: Optional[int] = None
    disable_by_batch_size: Optional[int] = None
    enable_suffix_decoding: bool = False
    suffix_cache_max_depth: int = 64
    suffix_max_spec_factor: float = 1.0
    suffix_max_spec_offset: float = 0.0
    suffix_min_token_prob: float = 0.1
    # this build's one extra key: where a row ends for the proposers (runner_logic.py: "reference" | "single_advance");
    # None = the library default ("single_advance" since r04; "reference" = the plugin's literal double count)
    proposal_indexing: Optional[str] = None

    def __post_init__(self):
        use_suffix = self.method == "suffix" or (self.method is None and self.enable_suffix_decoding)
        if (use_suffix or self.method == "arctic") and self.disable_by_batch_size is None:
            logger.info("Defaulting disable_by_batch_size to 64")
            self.disable_by_batch_size = 64
        if use_suffix:
            self.method = "suffix"
            self.enable_suffix_decoding = True
            self.num_speculative_tokens = self.suffix_cache_max_depth


_classes = None


def _config_classes():
    """ArcticParallelConfig / ArcticSpeculativeConfig (config.py:27-62), subclasses of vLLM's dataclasses."""
    global _classes
    if _classes is not None:
        return _classes
    import dataclasses

    from vllm.config import ParallelConfig, SpeculativeConfig

    @dataclasses.dataclass
    class ArcticParallelConfig(ParallelConfig):
        ulysses_sequence_parallel_size: int = 1
        enable_shift_parallel: bool = False
        shift_parallel_threshold: int = 512

        def __post_init__(self, *args, **kwargs):
            if self.enable_shift_parallel and self.ulysses_sequence_parallel_size == 1:
                raise ValueError("ulysses_sequence_parallel_size must be > 1 when enable_shift_parallel is True.")
            super().__post_init__(*args, **kwargs)

        @property
        def world_size(self) -> int:
            return self.pipeline_parallel_size * self.tensor_parallel_size * self.ulysses_sequence_parallel_size

        @world_size.setter
        def world_size(self, value: int) -> None:  # ParallelConfig assigns PP*TP; ignored (config.py:46-52)
            pass

    @dataclasses.dataclass
    class ArcticSpeculativeConfig(SpeculativeConfig):
        enable_suffix_decoding: bool = False
        suffix_cache_max_depth: int = 64
        suffix_max_spec_factor: float = 1.0
        suffix_max_spec_offset: float = 0.0
        suffix_min_token_prob: float = 0.1
        proposal_indexing: Optional[str] = None     # runner_logic.py; not a key of the reference

    _classes = (ArcticParallelConfig, ArcticSpeculativeConfig)
    return _classes


def arctic_parallel_config_class():
    return _config_classes()[0]


def build_config_patches():
    """ArcticPatch subclasses for vLLM's config classes; call only when vLLM is importable."""
    from vllm.config import ParallelConfig, SpeculativeConfig, VllmConfig
    from vllm.transformers_utils.configs.mlp_speculator import MLPSpeculatorConfig

    from ..patching import ArcticPatch

    ArcticParallelConfig, ArcticSpeculativeConfig = _config_classes()

    class ParallelConfigPatch(ArcticPatch[ParallelConfig]):
        def __new__(cls, *args, **kwargs):
            if cls is ParallelConfig:
                return ArcticParallelConfig.__new__(ArcticParallelConfig, *args, **kwargs)
            return super(ParallelConfig, cls).__new__(cls)

    class SpeculativeConfigPatch(ArcticPatch[SpeculativeConfig]):
        _orig_from_dict = SpeculativeConfig.__dict__["from_dict"].__wrapped__
        _orig_post_init = SpeculativeConfig.__post_init__

        def __new__(cls, *args, **kwargs):
            if cls is SpeculativeConfig:
                return ArcticSpeculativeConfig.__new__(ArcticSpeculativeConfig, *args, **kwargs)
            return super(SpeculativeConfig, cls).__new__(cls)

        def __post_init__(self):
            use_suffix = self.method == "suffix" or (self.method is None and self.enable_suffix_decoding)
            if (use_suffix or self.method == "arctic") and self.disable_by_batch_size is None:
                logger.info("Defaulting disable_by_batch_size to 64")
                self.disable_by_batch_size = 64
            if use_suffix:
                self.method = "suffix"
                self.enable_suffix_decoding = True
                self.num_speculative_tokens = self.suffix_cache_max_depth
                self._verify_args()
            else:
                self._orig_post_init()

        @classmethod
        def from_dict(cls, dict_value: dict):
            if cls is SpeculativeConfig:
                return SpeculativeConfigPatch._orig_from_dict(ArcticSpeculativeConfig, dict_value)
            return SpeculativeConfigPatch._orig_from_dict(cls, dict_value)

    class VllmConfigPatch(ArcticPatch[VllmConfig]):
        _orig_str = VllmConfig.__str__

        def __str__(self, *args, **kwargs):
            pc = self.parallel_config
            return (self._orig_str(*args, **kwargs) +
                    f", ulysses_sequence_parallel_size={pc.ulysses_sequence_parallel_size}"
                    f", enable_shift_parallel={pc.enable_shift_parallel}"
                    f", shift_parallel_threshold={pc.shift_parallel_threshold}")

    class MLPSpeculatorConfigPatch(ArcticPatch[MLPSpeculatorConfig]):
        _orig_init = MLPSpeculatorConfig.__init__

        def __init__(self, *args, **kwargs):
            self.base_model_arch = kwargs.pop("base_model_arch", "")     # config.py:128-133
            self._orig_init(*args, **kwargs)

    return [ParallelConfigPatch, SpeculativeConfigPatch, VllmConfigPatch, MLPSpeculatorConfigPatch]
