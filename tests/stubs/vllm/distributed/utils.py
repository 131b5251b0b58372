def get_pp_indices(num_hidden_layers: int, pp_rank: int, pp_size: int):
    per = num_hidden_layers // pp_size
    start = pp_rank * per
    end = num_hidden_layers if pp_rank == pp_size - 1 else start + per
    return start, end
