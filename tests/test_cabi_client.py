"""The C-ABI from plain C (tests/cabi_client.c, gcc -std=c11 -pedantic -Werror): the header is C, the library links
without Python / torch / C++, and one EM pass through it matches the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "cabi_client")


def _build():
    import __graft_entry__ as g
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(os.path.join(ROOT, "tests", "cabi_client.c")):
        g.build()
    assert os.path.exists(BIN)


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
