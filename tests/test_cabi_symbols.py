"""The C-ABI library loads on a CPU-only box and exports every symbol include/rbvfit_amd.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "rbvfit_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vp_[a-z_0-9]+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("vp_ctx_create", "vp_ctx_destroy", "vp_set_bounds", "vp_add_instrument", "vp_lnprob_batch",
                 "vp_lnprob_batch_device", "vp_model_flux_batch", "vp_last_error", "vp_device_count"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from rbvfit_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"librbvfit_amd.so does not export {name}"
        assert name in _lib.SIGNATURES, f"ctypes binding misses {name}"
    assert set(_lib.SIGNATURES) == set(declared_functions())
    assert b"rbvfit_amd" in lib.vp_version()


def test_one_version_string_everywhere():
    """Header macro, vp_version() of the built library and the Python package carry the same number."""
    import rbvfit_amd
    from rbvfit_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rbvfit_amd.h")).read()
    ver = re.search(r'#define\s+RBVFIT_AMD_VERSION\s+"([^"]+)"', hdr).group(1)
    assert rbvfit_amd.__version__ == ver
    assert _lib.load().vp_version().decode() == f"rbvfit_amd {ver} (gfx950, hip)"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from rbvfit_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RbvfitAmdLibraryError):
        _lib.load()


def test_no_gpu_is_an_error_not_a_fallback():
    """On a box without a GPU the product path raises; it never routes through a CPU path."""
    import rbvfit_amd
    if rbvfit_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rbvfit_amd.RbvfitAmdError):
        rbvfit_amd.Engine(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rbvfit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(("#", "//", "of the", "``")))
                assert not re.search(r"^\s*(import|from)\s+oracle", code, flags=re.M), f
                assert "voigt_oracle" not in src, f


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """Loading the library first and importing torch afterwards must not map a second HIP runtime
    (a torch wheel bundles its own libamdhip64.so; two copies leave torch with "No HIP GPUs")."""
    import subprocess
    import sys
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU box: no child processes from a test run that may have initialised the GPU")
    code = ("import rbvfit_amd\nfrom rbvfit_amd import _lib\n_lib.load()\nimport torch\n"
            "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
            "print(len(libs), libs)\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().startswith("1 "), out.stdout
