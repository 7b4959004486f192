"""CPU suite, part 2: the C-ABI library loads without a GPU and exports every symbol
include/mi355_spmv.h declares; argument checks that need no device; the Python
binding refuses host tensors (there is no CPU compute path)."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355_spmv.h")).read()
    names = set(re.findall(r"\b(mi355_spmv_[a-z_0-9]+)\s*\(", text))
    names -= {"mi355_spmv_plan", "mi355_spmv_dist", "mi355_spmv_plan_shape", "mi355_spmv_plan_info"}
    # the one-shot family is declared through a macro
    for kind in re.findall(r"MI355_SPMV_DECLARE_KIND\((\w+)\)", text):
        if kind == "KIND":
            continue
        for suf in ("i32_f32", "i32_f64", "i64_f32", "i64_f64"):
            names.add("mi355_spmv_%s_%s" % (kind, suf))
    for suf in re.findall(r"^MI355_SPMV_DECLARE_GENL\((\w+),", text, flags=re.M):
        names.add("mi355_spmv_merge_genl_%s" % suf)
    return {n for n in names if "##" not in n}


def test_library_exports_every_declared_symbol(sp):
    lib = sp.capi.lib()
    decl = declared_symbols()
    assert len(decl) >= 12 + 9
    missing = [s for s in sorted(decl) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(sp.capi.EXPORTS) == decl


def test_loader_library_exports_every_declared_symbol_and_loads_a_fixture(sp):
    """include/mi355_load.h (libmi355load.so, host only): exports == declarations; a golden fixture loads to the
    arrays tests/golden/golden.json holds (the reference's own LoadCoo + ToCsr made them, oracle/make_golden.py)."""
    import json
    text = open(os.path.join(ROOT, "include", "mi355_load.h")).read()
    decl = set(re.findall(r"\b(mi355_(?:load|csr_host)[A-Za-z_0-9]*)\s*\(", text))
    lib = sp.load.lib()
    assert decl == set(sp.load.EXPORTS) and all(hasattr(lib, s) for s in decl), decl
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    for name, case in gold.items():
        st = case["struct"]
        m = sp.load.load_mtx(os.path.join(ROOT, "tests", "golden", name))
        assert (m.n_rows, m.n_cols, m.nnz) == (st["n_rows"], st["n_cols"], st["nnz"]), name
        assert m.Ap.tolist() == st["Ap"] and m.Aj.tolist() == st["Aj"], name
    with pytest.raises(RuntimeError, match="could not be opened"):
        sp.load.load_mtx("/nonexistent/none.mtx")


def test_version_and_status_strings(sp):
    lib = sp.capi.lib()
    assert lib.mi355_spmv_version() == 310
    assert lib.mi355_spmv_status_string(0) == b"ok"
    assert lib.mi355_spmv_status_string(1) == b"invalid argument"
    assert lib.mi355_spmv_status_string(99) == b"unknown status"


def test_plan_create_rejects_bad_arguments_without_touching_the_device(sp):
    lib = sp.capi.lib()
    h = C.c_void_p()
    dummy = C.c_void_p(256)
    bad = [
        (7, 0, 0, 4, 4, 4, dummy, dummy),       # unknown kind
        (0, 5, 0, 4, 4, 4, dummy, dummy),       # unknown offset type
        (0, 0, 9, 4, 4, 4, dummy, dummy),       # unknown value type
        (0, 0, 0, -1, 4, 4, dummy, dummy),      # negative rows
        (0, 0, 0, 4, 4, -1, dummy, dummy),      # negative nnz
        (0, 0, 0, 4, 4, 2 ** 31, dummy, dummy), # nnz beyond 32-bit offsets
        (0, 0, 0, 4, 4, 4, None, dummy),        # null Ap
        (0, 0, 0, 4, 4, 4, dummy, None),        # null Aj
        (0, 0, 0, 4, 0, 4, dummy, dummy),       # nonzeros but no columns
    ]
    for kind, ot, vt, nr, nc, nnz, Ap, Aj in bad:
        st = lib.mi355_spmv_plan_create(C.byref(h), kind, ot, vt, nr, nc, nnz, Ap, Aj, 0)
        assert st == 1, (kind, ot, vt, nr, nc, nnz)
        assert not h.value
        assert lib.mi355_spmv_last_error() != b""
    assert lib.mi355_spmv_plan_execute(None, None, None, None, None) == 1
    assert lib.mi355_spmv_plan_destroy(None) == 0


def test_dist_entry_points_reject_bad_arguments_without_touching_the_device(sp):
    """The multi-GPU entry points (include/mi355_spmv.h, mi355_spmv_dist_*) check their arguments before
    any device or RCCL call."""
    lib = sp.capi.lib()
    h = C.c_void_p()
    dummy = C.c_void_p(256)
    assert lib.mi355_spmv_dist_create_local(C.byref(h), 0, 0, 0, 4, 4, 4, dummy, dummy, 0, None, 1, 0) == 1   # no devices
    assert lib.mi355_spmv_dist_create_local(C.byref(h), 0, 0, 0, 4, 4, 4, dummy, dummy, 1, None, 0, 0) == 1   # no sub-blocks
    assert lib.mi355_spmv_dist_create_local(None, 0, 0, 0, 4, 4, 4, dummy, dummy, 1, None, 1, 0) == 1
    cuts = (C.c_int64 * 3)(0, 4, 8)
    assert lib.mi355_spmv_dist_create_rank(C.byref(h), 0, 0, 0, 2, 2, None, 1, cuts, cuts, cuts, None, 4, 4, 4,
                                           dummy, dummy, 0) == 1                                              # rank >= world
    assert lib.mi355_spmv_dist_create_rank(C.byref(h), 0, 0, 0, 0, 2, None, 1, cuts, cuts, cuts, None, 4, 4, 4,
                                           dummy, dummy, 0) == 1                                              # world 2, no id
    assert not h.value
    assert lib.mi355_spmv_dist_execute(None, None, None, None, None) == 1
    assert lib.mi355_spmv_dist_destroy(None) == 0
    assert lib.mi355_spmv_dist_parts(None) == 0
    assert lib.mi355_spmv_knobs_reload() == 0


def test_binding_refuses_host_tensors_and_unknown_kinds(sp):
    Ap = torch.tensor([0, 1], dtype=torch.int32)
    Aj = torch.tensor([0], dtype=torch.int32)
    Ax = torch.ones(1)
    with pytest.raises(RuntimeError, match="device tensors only"):
        sp.spmv("vector", 1, 1, 1, Ap, Aj, Ax, Ax, Ax.clone())
    with pytest.raises(ValueError, match="NOT SUPPORTED"):
        sp.spmv("cusparse", 1, 1, 1, Ap, Aj, Ax, Ax, Ax.clone())


def test_missing_library_fails_loudly(sp, monkeypatch):
    monkeypatch.setattr(sp.capi, "_lib", None)
    monkeypatch.setattr(sp.capi, "LIB_PATH", "/nonexistent/libmi355spmv.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sp.capi.lib()
