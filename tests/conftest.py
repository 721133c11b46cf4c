import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


if os.environ.get("A3D_POISON", "0") == "1":
    # debugging aid: every torch.empty / empty_like of the run starts out as NaN (0xFF bytes for integers), and the
    # training tapes' shared scratch is re-poisoned at every use (agile3d_amd/backward.py) -- a kernel that reads memory
    # nothing wrote turns the results into NaN instead of into a small, allocation-history-dependent error
    _empty, _empty_like = torch.empty, torch.empty_like

    def _poison(t):
        if t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype != torch.bool:
                if t.dim() == 0:               # (torch's DataLoader makes a 0-d int64: no byte view of that)
                    t.fill_(-1)
                elif t.is_contiguous():
                    t.view(torch.uint8).fill_(255)
        return t

    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def decoder_weights():
    z = np.load(os.path.join(GOLDEN, "decoder_weights.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def golden_cases():
    return sorted(f[len("decoder_case_"):-4] for f in os.listdir(GOLDEN) if f.startswith("decoder_case_"))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"decoder_case_{name}.npz"))
    return {k: z[k] for k in z.files}


def arrays_to_clicks(rows, objs, times, K):
    ci = {str(o): [] for o in range(K + 1)}
    ct = {str(o): [] for o in range(K + 1)}
    for r, o, t in zip(rows.tolist(), objs.tolist(), times.tolist()):
        ci[str(o)].append(int(r))
        ct[str(o)].append(int(t))
    return ci, ct


@pytest.fixture(scope="session")
def full_model_cpu():
    """Our parameter tree with seeded weights and randomised BN statistics (CPU)."""
    from agile3d_amd.model import build_model, default_args, randomize_bn_stats
    torch.manual_seed(0)
    return randomize_bn_stats(build_model(default_args())).eval()
