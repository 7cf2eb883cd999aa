import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    """One khg context for the whole GPU session (fails loudly without a GPU)."""
    from kaldi_hmm_gmm_amd import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture
def opt(ctx):
    """Set library options (khg_ctx_set_option) for one test; the previous values come back afterwards.
    opt("k3_form", 1) ...; opt.k1("pdf") selects a K1 form by name."""
    saved = []

    class Setter:
        def __call__(self, name, value):
            saved.append((name, ctx.set_option(name, int(value))))

        def k1(self, form):
            saved.append(("k1_form", ctx.get_option("k1_form")))
            ctx.set_k1_form(form)

    yield Setter()
    for name, old in reversed(saved):
        ctx.set_option(name, old)
