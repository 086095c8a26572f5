"""The driver's entry points must not go stale: run `__graft_entry__.smoke()` as part of the GPU suite."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.gpu
def test_graft_entry_smoke(capsys):
    import __graft_entry__ as g
    g.smoke()
    assert "smoke ok" in capsys.readouterr().out
