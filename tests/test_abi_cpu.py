"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950 without a GPU,
loads, exports every symbol include/bgamd.h declares, and refuses to run without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from backgammon_env import _capi
    return _capi


def test_header_symbols_all_exported(built):
    hdr = open(os.path.join(ROOT, "include", "bgamd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bgamd_\w+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in bgamd.h but not exported"
    assert declared == {n for n, _, _ in built.SYMBOLS}, "ctypes table and header out of sync"


def test_no_cpu_fallback(built):
    import torch
    import backgammon_env as bg
    lib = built.load()
    assert lib.bgamd_version() >= 100
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.bgamd_device_count() == 0
    h = ctypes.c_void_p()
    assert lib.bgamd_env_create(ctypes.byref(h), 64, 0, 1, 0, 0, 0) == -3      # BGAMD_E_NODEVICE
    with pytest.raises(bg.BgamdError):
        bg.VecGame(64)
    with pytest.raises(bg.BgamdError):
        bg.Game(0)


def test_product_never_touches_oracle():
    pkg = os.path.join(ROOT, "backgammon-engine_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in src.replace("oracle/bg_oracle.c, which holds the known answers", ""), f


def test_host_mirror_types():
    import backgammon_env as bg
    assert bg.PlayerType.PLAYER1 == 0 and bg.PlayerType.PLAYER2 == 1 and int(bg.PlayerType.PLAYER2) == 1
    p = bg.Player("White", bg.PlayerType.PLAYER1)
    assert p.getName() == "White" and p.getNum() == 0
    with pytest.raises(TypeError):
        bg.Player("x", 1)
    assert bg.ERR_MESSAGES[6] == "Invalid destination." and bg.ERR_MESSAGES[1] == "Invalid origin"
