class MessageQueue:
    def __init__(self, n_reader, n_local_reader, max_chunk_bytes=0):
        self.n_reader, self.max_chunk_bytes, self.ready = n_reader, max_chunk_bytes, False

    def export_handle(self):
        return ("handle", self.n_reader)

    def wait_until_ready(self):
        self.ready = True
