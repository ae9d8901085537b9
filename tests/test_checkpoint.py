"""Reference checkpoint layout: NNN.npz as objax.io.save_var_collection writes it + meta.npy,
read back the way experiments/regression/test.py:38-53,89-130 does.  Host-side tests; the GPU round trip is in
test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write_reference_style_run(d, names, values, args, index=7):
    """What a reference run leaves behind: objax names, raw tensors, pickled args."""
    np.savez(os.path.join(d, "%03d.npz" % index), names=np.array(names), **{str(i): v for i, v in enumerate(values)})
    np.save(os.path.join(d, "meta.npy"), dict(args=args))


def test_read_run_matches_by_last_component_with_fallbacks(tmp_path):
    from smnngp import checkpoint as CK
    d = str(tmp_path)
    names = ["(SPR).kernel(NNGPKernel).w_std", "(SPR).kernel(NNGPKernel).b_std", "(SPR).likelihood(StudentTLikelihood).a",
             "(SPR).likelihood(StudentTLikelihood).b", "(SPR).diag_reg"]
    vals = [np.array(0.3, np.float32), np.array(-1.2, np.float32), np.array(0.9, np.float32), np.array(1.7, np.float32),
            np.array(-4.0, np.float32)]
    args = dict(method="tp", network=None, num_hiddens=2, activation="relu", data_name="yacht", last_w_std=1.5)
    _write_reference_style_run(d, names, vals, args, index=7)
    _write_reference_style_run(d, names, [v + 1 for v in vals], args, index=3)       # older file: must not be picked
    assert CK.latest_index(d) == 7
    raw, ctx = CK.read_run(d)
    assert ctx["method"] == "tp" and ctx["network"] is None
    assert float(raw["w_std"]) == pytest.approx(0.3) and float(raw["b_std"]) == pytest.approx(-1.2)
    assert float(raw["a"]) == pytest.approx(0.9) and float(raw["b"]) == pytest.approx(1.7)
    assert float(raw["eps"]) == pytest.approx(-4.0)                                  # stored as diag_reg
    assert float(raw["last_w_std"]) == 1.5                                           # not stored: from the run's args
    raw3, _ = CK.read_run(d, 3)
    assert float(raw3["w_std"]) == pytest.approx(1.3)
    saved = CK.load_var_collection(os.path.join(d, "007.npz"))
    assert CK.get_from_vars(saved, "nope") is None
    with pytest.raises(FileNotFoundError):
        CK.latest_index(str(tmp_path / "missing"))


def test_module_names_follow_objax_scoping_and_round_trip(tmp_path):
    from smnngp import checkpoint as CK
    from smnngp.spax.base import ConstraintTrainVar, Module
    from smnngp.spax.bijectors import positive

    class Inner(Module):
        def __init__(self):
            self.w_std = ConstraintTrainVar(1.3, constraint=positive())
            self.b_std = ConstraintTrainVar(0.2, constraint=positive())

    class Outer(Module):
        def __init__(self):
            self.kernel = Inner()
            self.eps = ConstraintTrainVar(1e-3, constraint=positive())
            self.alias = self.eps                                                    # shared variable: stored once

    m = Outer()
    vc = m.vars()
    assert set(vc) == {"(Outer).kernel(Inner).w_std", "(Outer).kernel(Inner).b_std", "(Outer).eps", "(Outer).alias"}
    path = os.path.join(str(tmp_path), "001.npz")
    CK.save_var_collection(path, vc)
    saved = CK.load_var_collection(path)
    assert len(saved["names"]) == 3 and os.path.exists(path)                         # exact name, no ".npz.npz"
    for key, var in (("w_std", m.kernel.w_std), ("b_std", m.kernel.b_std), ("eps", m.eps)):
        assert float(CK.get_from_vars(saved, key)) == float(var.value)               # RAW values on disk
    assert m.kernel.w_std.safe_value == pytest.approx(1.3)


def test_checkpointer_keeps_recent_files_and_steps_on_best(tmp_path):
    from smnngp import checkpoint as CK
    from smnngp.spax.base import TrainVar
    d = str(tmp_path / "run")
    ck = CK.Checkpointer(d, keep_ckpts=2)
    vc = {"(M).x": TrainVar(1.0)}
    assert ck.step(1, 5.0, vc) and not ck.step(2, 6.0, vc) and ck.step(3, 4.0, vc) and ck.step(10, 3.0, vc)
    assert sorted(os.listdir(d)) == ["003.npz", "010.npz"]
    with pytest.raises(TypeError):
        ck.save(11, [1, 2])
    CK.save_meta(d, dict(method="gp", num_hiddens=1))
    assert CK.load_meta(d)["method"] == "gp"
