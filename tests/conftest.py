import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def separated():
    """Well-separated greedy runs of the reference (tests/golden/make_golden.py separated, section G9)."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "separated_vectors.npz"))


@pytest.fixture(scope="session")
def full_model():
    """The 1.7B architecture (28 layers, V = 151 936) with random-init weights (reference init semantics, lm_head x 8 as
    in bench.py), built once per GPU test session: (config, model on cuda)."""
    import torch
    from coda_amd import CoDAConfig, CoDALanguageModel
    cfg = CoDAConfig()
    assert (cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_size) == (28, 151936, 2048)
    torch.manual_seed(4321)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            m = CoDALanguageModel(cfg)
    finally:
        torch.set_default_dtype(old)
    m.eval()
    with torch.no_grad():
        m.lm_head.weight.mul_(8.0)
    return cfg, m


@pytest.fixture(scope="session", autouse=True)
def _progress_heartbeat():
    """The full-size parity tests spend minutes inside the CPU oracle without printing; the GPU box takes a command that
    writes nothing for three minutes to be hung.  A daemon thread notes the elapsed time in gpurun_out/heartbeat.log once a
    minute while the session runs (only where a GPU is present: the CPU suite has no such test)."""
    import threading
    import time
    import torch
    if not torch.cuda.is_available():
        yield
        return
    stop = threading.Event()
    path = os.path.join(ROOT, "gpurun_out", "heartbeat.log")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    t0 = time.time()

    def beat():
        while not stop.wait(60.0):
            with open(path, "a") as f:
                f.write(f"pytest session alive, {time.time() - t0:.0f} s\n")
    th = threading.Thread(target=beat, daemon=True)
    th.start()
    yield
    stop.set()
