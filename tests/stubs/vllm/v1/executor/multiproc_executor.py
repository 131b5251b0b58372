from dataclasses import dataclass
from typing import Any


@dataclass
class UnreadyWorkerProcHandle:
    proc: Any
    rank: int


class _MQ:
    def wait_until_ready(self):
        self.ready = True


@dataclass
class WorkerProcHandle:
    proc: Any
    rank: int
    worker_response_mq: Any


class WorkerProc:
    made = []

    @staticmethod
    def make_worker_process(vllm_config, local_rank, rank, distributed_init_method, input_shm_handle):
        WorkerProc.made.append((local_rank, rank, distributed_init_method, input_shm_handle))
        return UnreadyWorkerProcHandle(proc=("proc", rank), rank=rank)

    @staticmethod
    def wait_for_ready(unready):
        return [WorkerProcHandle(u.proc, u.rank, _MQ()) for u in unready]

    def shutdown(self):
        self.shut = "orig"


class MultiprocExecutor:
    def __init__(self, vllm_config):
        self.vllm_config = vllm_config
        self.parallel_config = vllm_config.parallel_config
        self.max_concurrent_batches = vllm_config.parallel_config.pipeline_parallel_size
        self.monitor_started = False
        self._init_executor()

    def _init_executor(self) -> None:
        self.world_size = self.parallel_config.world_size
        self.workers = []

    def shutdown(self):
        pass

    def start_worker_monitor(self):
        self.monitor_started = True

    def _ensure_worker_termination(self, procs):
        self.terminated = procs

    def _get_output_rank(self) -> int:
        return self.world_size - self.parallel_config.tensor_parallel_size
