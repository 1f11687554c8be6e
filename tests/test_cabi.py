"""The C-ABI library loads and exports every symbol include/mpcrl.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mpcrl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcrl_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for s in ["mpcrl_create", "mpcrl_destroy", "mpcrl_set_theta", "mpcrl_set_gamma", "mpcrl_reset", "mpcrl_solve",
              "mpcrl_get_iterate", "mpcrl_set_iterate"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build()
    lib = ctypes.CDLL(g.LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/mpcrl.h but not exported"
    lib.mpcrl_version.restype = ctypes.c_int
    assert lib.mpcrl_version() >= 100


def test_python_binding_matches_header():
    from mpc4rl_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()
    # struct layout: field order of MpcrlProblemSpec
    hdr = open(os.path.join(ROOT, "include", "mpcrl.h")).read()
    body = hdr[hdr.index("typedef struct {"): hdr.index("} MpcrlProblemSpec;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.replace("typedef struct {", "").strip()
        if not stmt:
            continue
        for part in stmt.split(","):
            names.append(re.sub(r"[\*\s]", " ", part).split()[-1])
    assert names == [f[0] for f in _lib.ProblemSpec._fields_]


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mpc4rl_amd/ or bench's product path may import it."""
    pkg = os.path.join(ROOT, "mpc4rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mpc4rl_amd import MPCBatch, cartpole_ocp
    with pytest.raises(RuntimeError):
        MPCBatch(cartpole_ocp(), 4)
