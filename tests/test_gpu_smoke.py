import pytest


@pytest.mark.gpu
def test_smoke_entry():
    from smoke_impl import run_smoke
    assert run_smoke() <= 1e-3
