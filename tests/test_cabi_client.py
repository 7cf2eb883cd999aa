"""The C-ABI from plain C (tests/cabi_client.c, gcc -std=c11 -pedantic -Werror): the header is C, the library links
without Python / torch / C++, and one EM pass through it matches the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "cabi_client")


HOST_BIN = os.path.join(ROOT, "tests", "_build", "host_client")


def _build():
    import __graft_entry__ as g
    csrc = os.path.join(ROOT, "kaldi_hmm_gmm_amd", "csrc")
    hdr = os.path.join(ROOT, "include", "khg_hip.h")
    host = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.startswith("khg_host_") and f.endswith((".cpp", ".hpp"))]

    def stale(b, srcs):         # the binaries compile the host classes in: an edit to any of them (or to the header) rebuilds
        return not os.path.exists(b) or any(os.path.getmtime(b) < os.path.getmtime(s) for s in srcs)
    if stale(BIN, [os.path.join(ROOT, "tests", "cabi_client.c"), hdr]) or stale(HOST_BIN, [os.path.join(ROOT, "tests", "host_client.cpp"), hdr] + host):
        g.build()
    assert os.path.exists(BIN) and os.path.exists(HOST_BIN)


def test_c_client_compiles_and_host_entry_points_run():
    _build()
    r = subprocess.run([BIN, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "CABI_CLIENT_OK" in r.stdout or "CABI_CLIENT_SKIP" in r.stdout


@pytest.mark.gpu
def test_c_client_em_pass_matches_oracle():
    _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "CABI_CLIENT_OK" in r.stdout


def test_cpp_host_classes_without_python():
    """tests/host_client.cpp: HmmTopology, TransitionModel (+ MleUpdate), AmDiagGmm, AccumAmDiagGmm, MleAmDiagGmmUpdate,
    MapAmDiagGmmUpdate, StdVectorFst, AddTransitionProbs, ModifyGraphForCarefulAlignment from a plain C++ program linked against
    libkhg_hip.so -- the host classes carry no pybind11 / Python dependency."""
    _build()
    r = subprocess.run([HOST_BIN, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "HOST_CLIENT_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_host_classes_score_accumulate_and_align_on_the_gpu():
    _build()
    r = subprocess.run([HOST_BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HOST_CLIENT_OK" in r.stdout, r.stdout + r.stderr
