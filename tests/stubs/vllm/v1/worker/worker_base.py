class WorkerBase:
    def __init__(self, vllm_config, local_rank: int = 0, rank: int = 0, distributed_init_method: str = "", is_driver_worker: bool = False):
        self.vllm_config, self.local_rank, self.rank = vllm_config, local_rank, rank
