import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _dirty_device_memory():
    """FSQ_TEST_DIRTY_ALLOC=1: every CUDA tensor torch.empty hands out during the session is pre-filled with random bits - what a
    long-lived process gets from the caching allocator instead of the zero pages of a fresh one (round 3's fuzz found a kernel that
    only survived on the latter).  Off by default (it slows the suite down); the GPU box run of round 3 passed with it on."""
    if os.environ.get("FSQ_TEST_DIRTY_ALLOC") != "1":
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    real_empty = torch.empty
    gen = torch.Generator(device="cuda").manual_seed(12345)

    noise = {}

    def dirty_empty(*a, **k):
        t = real_empty(*a, **k)
        if t.is_cuda and t.numel() and t.is_contiguous() and t.dtype != torch.bool:
            raw = t.view(torch.uint8).reshape(-1)
            if t.device not in noise:       # (one 64 MiB block of random bits, copied in at a random phase: fast also for 35 GB workspaces)
                noise[t.device] = torch.randint(0, 256, (1 << 26,), dtype=torch.uint8, device=t.device, generator=gen)
            nz = noise[t.device]
            off = int(torch.randint(0, 1 << 20, (1,)).item())
            for a0 in range(0, raw.numel(), nz.numel() - (1 << 20)):
                n = min(nz.numel() - (1 << 20), raw.numel() - a0)
                raw[a0:a0 + n].copy_(nz[off:off + n])
            torch.cuda.current_stream(t.device).synchronize()      # (the fill must not race with the buffer's first use on another stream)
        return t
    torch.empty = dirty_empty
    try:
        yield
    finally:
        torch.empty = real_empty
