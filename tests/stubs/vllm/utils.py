import argparse
import importlib
import socket


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def get_open_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def get_distributed_init_method(ip: str, port: int) -> str:
    return f"tcp://{ip}:{port}"


def resolve_obj_by_qualname(qualname: str):
    mod, name = qualname.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


class FlexibleArgumentParser(argparse.ArgumentParser):
    pass
