"""CPU-side checks of the boundary: the C-ABI library loads without a GPU and exports every symbol
include/smnngp.h declares; the ctypes table covers exactly those symbols; no compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smnngp.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(smn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from smnngp import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libsmnngp.so does not export %s" % n
    assert sorted(lib.PROTOTYPES) == names, set(lib.PROTOTYPES) ^ set(names)


def test_prototype_arity_matches_header(lib):
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, args in lib.PROTOTYPES.items():
        m = re.search(r"\bint\s+%s\s*\((.*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        body = m.group(1).strip()
        n = 0 if body in ("", "void") else len(body.split(","))
        assert n == len(args), (name, n, len(args))


def test_version_and_no_device_fails_loudly(lib):
    assert lib._lib.smn_version() >= 100
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(lib.SmnError):
        lib.Context()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "scale-mixtures-of-neural-network-gaussian-processes_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                text = open(os.path.join(dp, f)).read()
                assert "oracle" not in text.lower() or f == "build.py", os.path.join(dp, f)


def test_lazy_device_array_algebra_is_symbolic(lib):
    j = lib.ScaledIdentity(4, 0.5)
    assert np.allclose(np.asarray(2.0 * j), np.eye(4))
    from smnngp.spax.utils import jitter
    assert isinstance(jitter(3, 1e-6), lib.ScaledIdentity) and jitter(3).eps == 1e-6


def test_bijectors_and_trainvars():
    from smnngp.spax.base import ConstraintTrainVar
    from smnngp.spax.bijectors import positive
    for v in (1e-8, 1e-6, 0.3, 1.0, 19.0, 25.0):
        assert abs(ConstraintTrainVar(v, positive()).safe_value - v) <= 1e-9 * max(v, 1.0) + 1e-16
    assert ConstraintTrainVar(25.0, positive()).value == 25.0        # bijectors.py:53 guard
    assert abs(ConstraintTrainVar(0.7, positive(base="exp")).safe_value - 0.7) < 1e-12
    with pytest.raises(NotImplementedError):
        from smnngp.spax.bijectors import triangular
        triangular()


def test_digamma_and_bijector_derivatives():
    """Host pieces of the analytic gradient (SPR.loss_and_grad): psi against scipy, d constrained / d raw against
    central differences."""
    from scipy.special import digamma as ref
    from smnngp.spax.bijectors import positive
    from smnngp.spax.utils import digamma
    for x in (0.05, 0.5, 1.0, 2.5, 9.99, 10.0, 123.4, 8200.5):
        assert abs(digamma(x) - ref(x)) < 1e-12 * max(1.0, abs(ref(x)))
    with pytest.raises(ValueError):
        digamma(0.0)
    for base in ("softplus", "exp"):
        p = positive(base=base)
        for raw in (-3.0, 0.0, 0.7, 25.0):
            h = 1e-6
            fd = (p(raw + h) - p(raw - h)) / (2 * h)
            assert abs(p.grad(raw) - fd) < 1e-8 * max(1.0, abs(fd))



def test_plateau_schedule_follows_the_reference_semantics():
    """train.PlateauSchedule against the behaviour of the scheduler the regression run uses (experiments/utils.py:153-231):
    relative / absolute thresholds, min / max modes, patience counting, the floor and the eps guard."""
    from smnngp.train import PlateauSchedule
    s = PlateauSchedule(0.1, factor=0.5, patience=2)
    out = [s.step(v) for v in (1.0, 0.99999, 0.9, 0.91, 0.92, 0.93, 0.5)]
    #        best=1   no (rel 1e-4)  better  bad1  bad2  bad3>patience -> cut   better
    assert out == [False, False, False, False, False, True, False] and abs(s.lr - 0.05) < 1e-15 and s.best == 0.5
    m = PlateauSchedule(1.0, mode="max", factor=0.1, patience=0, threshold=0.1, threshold_mode="abs", min_lr=0.05)
    assert [m.step(v) for v in (1.0, 1.05, 1.2, 1.2)] == [False, True, False, True] and abs(m.lr - 0.05) < 1e-15
    assert m.step(0.0) and m.lr == 0.05                                  # already at the floor: reported, not lowered
    inf = PlateauSchedule(1e-9, factor=0.5, patience=0, eps=1e-8)
    inf.step(1.0); assert inf.step(2.0) and inf.lr == 1e-9               # a cut smaller than eps is not applied
    with pytest.raises(ValueError):
        PlateauSchedule(0.1, mode="down")
    with pytest.raises(ValueError):
        PlateauSchedule(0.1, threshold_mode="pct")


def test_generated_panel_leaf_include_matches_its_generator():
    """csrc/panel_leaf_steps.inc is inline asm with hand-placed wait states: the committed file must be exactly
    what gen_panel_leaf.py writes, so an edit of one without the other cannot ship."""
    import importlib.util
    csrc = os.path.join(ROOT, "scale-mixtures-of-neural-network-gaussian-processes_amd", "csrc")
    spec = importlib.util.spec_from_file_location("gen_panel_leaf", os.path.join(csrc, "gen_panel_leaf.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(csrc, "panel_leaf_steps.inc")) as f:
        assert f.read() == mod.render()


def test_clean_tree_build_from_scratch(tmp_path):
    """`build()` is mtime-incremental and the built objects travel in the working tree, so the everyday build check can
    pass on stale objects.  This one compiles EVERY source from scratch (force=True, a variant directory of its own, removed
    afterwards) and checks that the result exports every declared symbol."""
    import importlib.util
    import shutil
    pkg = os.path.join(ROOT, "scale-mixtures-of-neural-network-gaussian-processes_amd")
    spec = importlib.util.spec_from_file_location("smnngp_build_clean", os.path.join(pkg, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    objdir = os.path.join(pkg, "build_cleancheck")
    lib_path = os.path.join(pkg, "libsmnngp_cleancheck.so")
    shutil.rmtree(objdir, ignore_errors=True)
    if os.path.exists(lib_path):
        os.unlink(lib_path)
    try:
        out = mod.build(force=True, variant="cleancheck")
        assert out == lib_path and os.path.exists(lib_path)
        assert sorted(f for f in os.listdir(objdir) if f.endswith(".o")) == sorted(s.replace(".hip", ".o") for s in mod.SOURCES)
        raw = ctypes.CDLL(lib_path)
        for n in declared_symbols():
            assert hasattr(raw, n), n
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
        if os.path.exists(lib_path):
            os.unlink(lib_path)
