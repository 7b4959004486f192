"""The MI355X harness (spmv-samples_amd/host/main.cpp): argv shape, error
behaviour of the loader it embeds (CPU), and the two tables of main.cu:83-113 (GPU)."""
import os
import re
import subprocess

import pytest

from conftest import GOLD, ROOT

EXE = os.path.join(ROOT, "spmv-samples_amd", "bin", "spmv")


@pytest.fixture(scope="module")
def exe():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "spmv-samples_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "spmv-samples_amd", "host")], check=True)
    return EXE


def run(exe, *args):
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=600)


def test_usage_message_and_exit_code(exe):
    """main.cu:22-25."""
    r = run(exe)
    assert r.returncode == 1 and "usage: ./bin/<program-name>  <filename.mtx>  <SpMV_kind_string>..." in r.stderr
    r = run(exe, "only_a_file.mtx")
    assert r.returncode == 1 and "usage:" in r.stderr


@pytest.mark.parametrize("text,msg", [
    (None, "File could not be opened"),                                                     # load.hpp:278-281
    ("%%NotMM matrix coordinate real general\n1 1 1\n1 1 1\n", "Could not process Matrix Market banner"),
    ("%%MatrixMarket matrix array real general\n1 1\n1\n", "File is not a sparse matrix"),    # load.hpp:289-292
    ("%%MatrixMarket matrix coordinate real general\n% c\n", "Could not read file info (M, N, NNZ)"),
    ("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n", "Unrecognized matrix market format type"),
])
def test_loader_failures_exit_like_the_reference(exe, tmp_path, text, msg):
    p = tmp_path / "m.mtx"
    if text is not None:
        p.write_text(text)
    r = run(exe, str(p), "hip_vector")
    assert r.returncode == 1
    assert msg in r.stderr


def test_malformed_entry_terminates_with_the_reference_message(exe, tmp_path):
    """An uncaught exception_t, as in the reference (load.hpp:324-329): abnormal termination."""
    p = tmp_path / "m.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n0 1 1.0\n")
    r = run(exe, str(p), "hip_vector")
    assert r.returncode != 0 and "Market file is zero-indexed" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--dtype", "f64"], ["--offset", "64"], ["--dtype", "f64", "--offset", "64"]])
def test_tables_match_the_reference_format(exe, extra):
    path = os.path.join(GOLD, "c1_1138_bus_standin.mtx")
    kinds = ["hip_vector", "hip_merge", "hip_light", "hip_auto", "hip_functor"]
    r = run(exe, path, *kinds, "--iters", "20", *extra)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Dataset: c1_1138_bus_standin.mtx\n\tn_rows: 1138  n_cols: 1138  nnz: 4054\n" in out   # main.cu:38-39
    assert "Compute delta:\n" in out and "\nTime cost:\n" in out
    for k in kinds:
        m = re.search(r"^\[%-12s\] sum: +([0-9.eE+-]+|nan)  avg: +([0-9.eE+-]+|nan)$" % k, out, re.M)
        assert m, out
        # values of magnitude ~1e3 with 3-4 entries per row: fp32 sums differ from the serial
        # order by a few ulp per row at most; NaN (a skipped row, y is poisoned) fails here
        assert float(m.group(2)) < (1e-3 if "f64" not in extra else 1e-10)
        t = re.search(r"^\[%-12s\] total: +([0-9.]+) ms  kernel: +([0-9.]+) ms$" % k, out, re.M)
        assert t, out
        assert float(t.group(2)) <= float(t.group(1))


@pytest.mark.gpu
def test_unknown_kind_and_unit_flag(exe):
    path = os.path.join(GOLD, "lattice9_cub_doc.mtx")
    r = run(exe, path, "hip_vector", "--iters", "5", "--unit-us")
    assert r.returncode == 0 and re.search(r"total: +[0-9.]+ us  kernel: +[0-9.]+ us", r.stdout)
    assert re.search(r"\[hip_vector  \] sum: +0\.000000  avg: +0\.000000", r.stdout)       # integer-valued: exact
    r = run(exe, path, "no_such_kind")
    assert r.returncode == 1 and 'SpMV kind "no_such_kind" is NOT SUPPROT' in r.stderr    # spmv.h:46-47


@pytest.mark.gpu
def test_harness_on_a_generated_banded_file(exe, tmp_path):
    """End to end on a file large enough for the parallel parser to cut it into many chunks
    (40 000 rows x 32 nonzeros, ~21 MB of text): loader -> CSR -> three kinds -> delta table."""
    import numpy as np
    n, per_row, w = 40000, 32, 600
    rng = np.random.RandomState(12)
    rows = np.repeat(np.arange(n), per_row)
    lo = np.clip(np.arange(n) - w, 0, n - 2 * w - 1)
    cols = (lo[:, None] + np.sort(rng.randint(0, 2 * w, size=(n, per_row)), axis=1)).reshape(-1)
    vals = rng.randint(-4, 5, size=n * per_row)            # small integers: every kind must be exact
    path = tmp_path / "band.mtx"
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate integer general\n%d %d %d\n" % (n, n, n * per_row))
        np.savetxt(f, np.stack([rows + 1, cols + 1, vals], axis=1), fmt="%d")
    r = run(exe, str(path), "hip_vector", "hip_merge", "hip_light", "--iters", "20")
    assert r.returncode == 0, r.stderr
    assert "n_rows: 40000  n_cols: 40000  nnz: 1280000" in r.stdout
    for k in ("hip_vector", "hip_merge", "hip_light"):
        assert re.search(r"^\[%-12s\] sum: +0\.000000  avg: +0\.000000$" % k, r.stdout, re.M), r.stdout


@pytest.mark.gpu
def test_the_references_own_labels_run_unchanged(exe):
    """A command line written for the reference (README.md:17-19: `./bin/spmv file.mtx cusp cusp1 ... merge_genl`)
    runs as it is: every label of spmv.h:18-27 except the vendor column names the MI355X kind standing in its place,
    and prints under the label it was asked by."""
    path = os.path.join(GOLD, "c1_1138_bus_standin.mtx")
    kinds = ["cusp", "cusp1", "cusp2", "light_vec", "light_warp", "cub_merge", "merge", "merge_genl"]
    r = run(exe, path, *kinds, "--iters", "5")
    assert r.returncode == 0, r.stderr
    for k in kinds:
        m = re.search(r"^\[%-12s\] sum: +([0-9.eE+-]+|nan)  avg: +([0-9.eE+-]+|nan)$" % k, r.stdout, re.M)
        assert m, r.stdout
        assert float(m.group(2)) < 1e-3
        assert re.search(r"^\[%-12s\] total: +([0-9.]+) ms  kernel: +([0-9.]+) ms$" % k, r.stdout, re.M), r.stdout
    # an unknown label: the reference's message and exit code (spmv.h:46-47)
    r = run(exe, path, "cusp3")
    assert r.returncode == 1 and 'SpMV kind "cusp3" is NOT SUPPROT' in r.stderr


@pytest.mark.gpu
def test_vendor_comparison_kinds_in_the_harness(exe):
    """`rocsparse` / `rocsparse_stream` (host/spmv/rocsparse_cmp.hpp): the place of `cusparse` in the reference's
    SPMV_KINDS (spmv.h:19), compiled into the harness only when rocSPARSE is there.  Same tables, same delta
    check against the CPU path; with 64-bit offsets the kind reports and exits (csrmv has 32-bit offsets)."""
    path = os.path.join(GOLD, "c1_1138_bus_standin.mtx")
    kinds = ["hip_vector", "rocsparse", "rocsparse_stream"]
    r = run(exe, path, *kinds, "--iters", "5")
    if "is NOT SUPPROT" in r.stderr:
        pytest.skip("harness built without rocSPARSE")
    assert r.returncode == 0, r.stderr
    for k in kinds:
        m = re.search(r"^\[%-12s\] sum: +([0-9.eE+-]+|nan)  avg: +([0-9.eE+-]+|nan)$" % k, r.stdout, re.M)
        assert m, r.stdout
        assert float(m.group(2)) < 1e-3
    r = run(exe, path, "rocsparse", "--offset", "64")
    assert r.returncode == 1 and "32-bit row offsets" in r.stderr


def test_synthetic_spec_errors_without_a_gpu(exe):
    """--synthetic / synthetic:<spec> in the file's place (SURVEY §5 "config / flags"): a bad spec is refused before
    anything touches the device."""
    r = run(exe, "--synthetic")
    assert r.returncode == 1 and "--synthetic needs a spec" in r.stderr
    r = run(exe, "synthetic:blob:n=10", "hip_vector")
    assert r.returncode == 1 and "family is band or rand" in r.stderr
    r = run(exe, "--synthetic", "band:n=100,k=300,w=10", "hip_vector")
    assert r.returncode == 1 and "synthetic spec: need" in r.stderr


@pytest.mark.gpu
def test_harness_on_seeded_synthetic_matrices(exe):
    """The harness on a generated matrix instead of a file: the delta table against the serial CPU loop, the same
    dataset under the same seed, another one under another seed."""
    outs = []
    for args in (["--synthetic", "band:n=200000,k=32,w=4096", "hip_vector", "hip_merge", "hip_light", "--iters", "10"],
                 ["synthetic:band:n=200000,k=32,w=4096", "hip_vector", "hip_merge", "hip_light", "--iters", "10", "--seed", "1"],
                 ["synthetic:band:n=200000,k=32,w=4096", "hip_vector", "hip_merge", "hip_light", "--iters", "10", "--seed", "9"],
                 ["--synthetic", "rand:n=100000,k=16", "hip_merge", "hip_dist_vector", "--iters", "5", "--dtype", "f64", "--offset", "64"]):
        r = run(exe, *args)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
        n, k = (200000, 32) if "band" in " ".join(args) else (100000, 16)
        assert "n_rows: %d  n_cols: %d  nnz: %d" % (n, n, n * k) in r.stdout
        for m in re.finditer(r"^\[(\S+) *\] sum: +([0-9.eE+-]+|nan)  avg: +([0-9.eE+-]+|nan)$", r.stdout, re.M):
            assert float(m.group(3)) < 1e-4, r.stdout             # |y - y_cpu| per row: rounding only
        assert len(re.findall(r"^\[\S+ *\] sum:", r.stdout, re.M)) == len([a for a in args if a.startswith("hip_")])
    delta = lambda o: re.findall(r"sum: +(\S+)", o)
    assert delta(outs[0]) == delta(outs[1])                          # default seed is 1
    assert delta(outs[0]) != delta(outs[2])
