import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import build as obuild, oracle

    obuild.build()
    return oracle


@pytest.fixture(scope="session")
def hiplib():
    from sparse_amd.csrc import build as hbuild

    hbuild.build()
    from sparse_amd import _ffi

    return _ffi.lib()


def init_single_rank_group(backend="nccl", attempts=5):
    """`init_process_group` at world size 1 on a port that is free NOW - and again on another one if somebody took it between
    the probe and the store's listen (seen once on the GPU box: EADDRINUSE on the probed port, right after the tests that
    launch bench.py under torch.distributed.run)."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    last = None
    for _ in range(attempts):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
        s.close()
        try:
            kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=0, world_size=1, **kw)
            return
        except Exception as e:      # DistNetworkError (EADDRINUSE): try the next free port
            last = e
            if "EADDRINUSE" not in str(e) and "address already in use" not in str(e):
                raise
    raise last
