"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950 without a GPU,
loads, exports every symbol include/bgamd.h declares, and refuses to run without a device."""
import ctypes
import os
import re

import numpy as np

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


def test_library_carries_the_digest_of_its_sources(built):
    """The loaded libbgamd.so was compiled from the sources next to it: build() is content-based, the binding refuses
    a library whose digest differs (a stale .so travels with gpurun snapshots), and the file can be checked without
    loading it."""
    from backgammon_env import _srchash
    want = _srchash.source_hash()
    assert len(want) == 16
    assert _srchash.library_hash(built.LIB_PATH) == want
    assert built.load().bgamd_source_hash().decode() == want == built.source_hash()


def test_digest_is_reproducible_from_the_commit(built, tmp_path):
    """The digest a library carries must be reproducible from a checkout of the commit it was built from: every file that
    enters it is tracked by git (round 2 hashed a git-ignored compiler temporary that happened to sit in csrc/), and the
    digest of `git archive HEAD` equals source_hash() whenever the hashed files have no uncommitted changes."""
    import hashlib
    import subprocess
    import tarfile
    from backgammon_env import _srchash
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout (gpurun snapshots travel without .git)")
    files = _srchash.source_files()
    rel = [os.path.relpath(f, ROOT) for f in files]
    assert all(f.endswith((".h", ".hip")) for f in rel)
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "--error-unmatch"] + rel, capture_output=True, text=True)
    assert tracked.returncode == 0, "hashed but untracked: " + tracked.stderr
    listed = set(subprocess.check_output(["git", "-C", ROOT, "ls-files", "backgammon-engine_amd/csrc"], text=True).split())
    assert {r for r in rel if r.startswith("backgammon-engine_amd/csrc")} == {l for l in listed if l.endswith((".h", ".hip"))}
    if subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--"] + rel).returncode != 0:
        pytest.skip("hashed sources have uncommitted changes: the commit's digest is another one")
    tar = tmp_path / "head.tar"
    subprocess.check_call(["git", "-C", ROOT, "archive", "-o", str(tar), "HEAD"] + rel)
    with tarfile.open(tar) as t:
        t.extractall(tmp_path / "head")
    h = hashlib.sha256()
    for r in rel:                                              # the digest of _srchash.source_hash(), over the archived files
        h.update(os.path.basename(r).encode() + b"\0")
        h.update((tmp_path / "head" / r).read_bytes())
    assert h.hexdigest()[:16] == _srchash.source_hash()


def test_checkpoint_interop_against_the_reference(built):
    """SURVEY §8f row 2.  tests/golden/f2_interop.json was written in the build container by make_golden_r2.py with the
    UNMODIFIED reference: its checkpoint through this repo's loader, and a state_dict written by this repo's learner
    through the reference's strict load_state_dict and train._model_compatible (train.py:361-381).  Here, without the
    reference: the record says pass, the checkpoint it was made from is the one the weight fixture holds, and the
    build's writer still produces exactly the layout (names, order, shapes, dtypes) the reference's reader accepted."""
    import hashlib
    import io
    import json

    import numpy as np
    import torch
    from backgammon_env.learner import TDLambdaLearner
    from backgammon_env.policy import TDLGammonModel, flatten_state_dict
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "f2_interop.json")))
    for k in ("build_loader_reads_reference_pth", "reference_strict_load_of_build_pth",
              "reference_model_compatible_of_build_pth", "weights_survive_roundtrip_bit_exact"):
        assert rec[k] is True, k
    assert rec["build_state_dict_layout"] == rec["reference_state_dict_layout"]
    fix = np.fromfile(os.path.join(ROOT, "tests", "golden", "tdgammonNEW100k.f32"), dtype=np.float32)
    rng = np.random.RandomState(5)                            # the very weights make_golden_r2.py wrote and the reference read
    w = (fix + rng.normal(0, 0.01, fix.shape)).astype(np.float32)
    sd = TDLambdaLearner(w, device="cpu").state_dict()
    assert list(sd) == list(rec["reference_state_dict_layout"])
    for k, v in sd.items():
        assert list(v.shape) == rec["reference_state_dict_layout"][k]["shape"] and str(v.dtype) == rec["reference_state_dict_layout"][k]["dtype"]
    assert hashlib.sha256(w.tobytes()).hexdigest() == rec["build_written_weights_sha256"]   # the weights the reference read back bit-exactly
    # torch.save -> torch.load(weights_only=True) -> both of the build's readers
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    back = torch.load(buf, map_location="cpu", weights_only=True)
    assert np.array_equal(flatten_state_dict(back), w)
    m = TDLGammonModel()
    m.load_state_dict(back)
    assert np.array_equal(m._w, w) and list(m.state_dict()) == list(sd)


def test_pybind_module_has_the_reference_surface(built):
    """The pybind11 extension over the C ABI exports exactly the classes and methods of the reference's
    PYBIND11_MODULE(backgammon_env) (cppsrc/backgammon_bindings.cpp:41-94), with its enum / constructor behaviour, and
    refuses to run without a device.  In a process of its own (tests/pybind_driver.py): CPython keeps one extension module
    per name, and this process may load the reference's module of the same name (oracle/_ref) for the oracle tests."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pybind_driver.py"), "surface"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.split()
    assert lines[0] == "OK" and lines[1] == built.source_hash()


def test_weights_outside_the_f16_split_are_refused(built):
    """the root pass multiplies fc1.weight as f16 hi + f16 lo planes; a table with |w| >= 65 504 (hi = inf, lo = NaN), a weight the
    split cannot hold, or a non-finite value anywhere must be refused loudly (BGAMD_E_WEIGHTS = -8), host side, before anything is loaded.
    The reference checkpoint and tables of ordinary size pass."""
    lib = built.load()
    w = np.fromfile(os.path.join(ROOT, "tests", "golden", "tdgammonNEW100k.f32"), dtype=np.float32)

    def chk(x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return lib.bgamd_weights_check(x.ctypes.data_as(ctypes.c_void_p))
    assert chk(w) == 0
    rng = np.random.default_rng(1)
    assert chk((rng.standard_normal(25601) * 30).astype(np.float32)) == 0          # any ordinary table: 22 mantissa bits hold
    for idx, val in ((17, 65520.0), (17, -7.0e4), (198 * 5 + 3, np.inf), (0, np.nan), (25344 + 5, np.nan), (25600, np.inf)):
        bad = w.copy()
        bad[idx] = val
        assert chk(bad) == -8, (idx, val)
    ok = w.copy()
    ok[17] = 60000.0                                                               # inside the f16 range
    assert chk(ok) == 0
    big15 = w.copy()
    big15[197] = 15 * 60000.0                                                      # features 196/197 carry w / 15 (the count's 1/15 is folded in)
    assert chk(big15) == 0
    big15[197] = 15 * 70000.0
    assert chk(big15) == -8
    assert b"65504" in lib.bgamd_error_string(-8)
