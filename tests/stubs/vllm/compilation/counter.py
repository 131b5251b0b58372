class _Counter:
    num_gpu_runner_capture_triggers = 0
    num_piecewise_capturable_graphs_seen = 0


compilation_counter = _Counter()
