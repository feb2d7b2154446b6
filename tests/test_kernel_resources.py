"""Register / scratch / LDS budget of the hot kernel instances, read from the built library's code-object metadata.

The measured rates hang on occupancy: six waves per SIMD need <= 80 VGPRs, seven <= 72 (the tile and walker kernels sit at
67-71), a kernel that touches scratch at all measured 1-1.5 us slower per launch, and the build depends on two -mllvm flags
(-disable-machine-licm, -amdgpu-sched-strategy=iterative-ilp).  A toolchain bump that silently changes any of this should
fail HERE, on the CPU, not show up as a 5-25 % slower bench line.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rbvfit_amd", "lib", "librbvfit_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernel (demangled prefix) -> limits.  vgpr: upper bound (the occupancy step it must stay under); scratch: bytes of private
# segment (0 = none); lds: static group segment bytes (the tile / walker kernels take ALL their LDS dynamically: 0)
PINS = {
    # C1 headline: one launch per batch.  <= 80 VGPRs = six waves per SIMD = two 12-wave walkers per CU
    "void vp::walker_kernel<0, false, false, false, 0, false>(": dict(vgpr=80, scratch=0, lds=0),
    # ... its stretch-move, pre-armed and flux forms
    "void vp::walker_kernel<0, false, true, false, 0, false>(": dict(vgpr=80, scratch=0, lds=0),
    "void vp::walker_kernel<0, false, false, true, 0, false>(": dict(vgpr=80, scratch=0, lds=0),
    "void vp::walker_kernel<0, false, false, false, 1, false>(": dict(vgpr=80, scratch=32, lds=0),
    # ... and the split form (several workgroups of one-pass tiles per walker: up to 13 waves per workgroup, two per CU need
    # <= 72 VGPRs = seven waves per SIMD)
    "void vp::walker_kernel<0, false, false, false, 0, true>(": dict(vgpr=72, scratch=0, lds=0),
    "void vp::walker_kernel<0, false, true, false, 0, true>(": dict(vgpr=72, scratch=0, lds=0),
    "void vp::walker_kernel<0, false, false, true, 0, true>(": dict(vgpr=72, scratch=0, lds=0),
    # C2-C4: single-wave tiles, 7 waves per SIMD at <= 72
    "void vp::tile_kernel1<0, true>(": dict(vgpr=72, scratch=0, lds=0),
    "void vp::tile_kernel1<0, false>(": dict(vgpr=72, scratch=0, lds=0),
    "void vp::tile_kernel<0, 0, false, true, false>(": dict(vgpr=72, scratch=0, lds=0),
    "void vp::tile_kernel<0, 0, false, false, false>(": dict(vgpr=72, scratch=0, lds=0),
    # far-field expansions: 4 waves per SIMD at <= 128
    "void vp::farfield_kernel<6, false>(": dict(vgpr=112, scratch=0, lds=0),
    "void vp::farfield_kernel<9, true>(": dict(vgpr=112, scratch=0, lds=0),
    "void vp::finalize_kernel<true>(": dict(vgpr=40, scratch=0, lds=0),
}


def _metadata(tmp_path):
    """{demangled kernel name: {vgpr, sgpr, scratch, lds, vgpr_spill, sgpr_spill}} of the gfx950 code object in the library."""
    objdump, readelf = (os.path.join(LLVM, t) for t in ("llvm-objdump", "llvm-readelf"))
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or os.path.join(LLVM, "llvm-cxxfilt")
    if not all(os.path.exists(t) for t in (objdump, readelf, filt)):
        pytest.skip("llvm-objdump / llvm-readelf (/opt/rocm/lib/llvm/bin) or c++filt not found")
    if not os.path.exists(LIB):
        import __graft_entry__ as ge
        ge.build()
    lib = shutil.copy(LIB, tmp_path / "lib.so")                 # (--offloading writes the bundles next to its input)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    cos = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(cos) == 1, os.listdir(tmp_path)
    notes = subprocess.run([readelf, "--notes", str(tmp_path / cos[0])], check=True, capture_output=True, text=True).stdout
    out = {}
    # one YAML map per kernel; '.name:' sits between the numeric keys, so cut the text at '- .agpr_count' / '- .args'
    for blk in re.split(r"\n\s+- \.a", notes):
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m or ".vgpr_count" not in blk:
            continue
        g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
        out[m.group(1)] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                               lds=g("group_segment_fixed_size"), vgpr_spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"))
    names = list(out)
    dem = subprocess.run([filt], input="\n".join(names), check=True, capture_output=True, text=True).stdout.splitlines()
    return {d: out[n] for n, d in zip(names, dem)}


def test_hot_kernel_instances_keep_their_register_and_scratch_budget(tmp_path):
    meta = _metadata(tmp_path)
    assert len(meta) > 40, "kernel metadata not found in the code object"
    report = []
    for prefix, lim in PINS.items():
        hits = [k for k in meta if k.startswith(prefix)]
        assert len(hits) == 1, (prefix, hits)
        m = meta[hits[0]]
        report.append(f"{prefix[5:-1]:55s} vgpr {m['vgpr']:3d} (<= {lim['vgpr']})  sgpr {m['sgpr']:3d}  scratch {m['scratch']:3d} B (<= {lim['scratch']})  "
                      f"spills v/s {m['vgpr_spill']}/{m['sgpr_spill']}")
        assert m["vgpr"] <= lim["vgpr"], report[-1]
        assert m["scratch"] <= lim["scratch"], report[-1]
        assert m["lds"] == lim["lds"], report[-1]
        if lim["scratch"] == 0:
            assert m["vgpr_spill"] == 0, report[-1]
    print("\n".join(report))
