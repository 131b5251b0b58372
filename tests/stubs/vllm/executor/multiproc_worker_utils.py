calls = []


def set_multiprocessing_worker_envs(parallel_config) -> None:
    calls.append(parallel_config)
