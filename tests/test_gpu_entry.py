"""The driver's entry points must keep working: __graft_entry__.smoke() is what runs on the GPU box at round end."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_smoke_entry_point(capsys):
    sys.path.insert(0, ROOT)
    entry = importlib.import_module("__graft_entry__")
    entry.smoke()
    out = capsys.readouterr().out
    assert out.count("smoke ok") == 5
