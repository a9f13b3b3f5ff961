"""Build-time guard of the band kernel (CPU container: hipcc cross-compiles gfx950).  The kernel's loader is the LDS-DMA path, whose only wait
is `s_waitcnt vmcnt(0)` in front of the plane barrier: a register spill reloaded inside the plane loop is a vector memory operation on the same
counter and would drain the DMA every plane (hipcc's allocation is erratic near the 64-VGPR cap: round 3 saw single-line edits move spills into
the loop).  So: no scratch operation anywhere in any instance, 64 VGPRs at most (8 waves per SIMD), the LDS budget of two workgroups per CU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason="needs hipcc")
def test_band_kernel_has_no_scratch_and_fits_two_workgroups_per_cu(tmp_path):
    src = os.path.join(ROOT, "ml-gmpi_amd", "csrc", "render_band.hip")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
             "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src)]   # = ml-gmpi_amd/csrc/Makefile
    mk = open(os.path.join(os.path.dirname(src), "Makefile")).read()
    for f in ("-ffp-contract=off", "-fno-slp-vectorize", "-O3"):
        assert f in mk, f"the Makefile no longer passes {f}: keep this test's flags in step with it"
    subprocess.run([HIPCC, *flags, "-save-temps", "-c", src, "-o", "render_band.o"], cwd=tmp_path, check=True, capture_output=True, timeout=900)
    asm = open(os.path.join(tmp_path, "render_band-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    names = sorted(set(re.findall(r"^(_ZN4gmpi4band18render_band_kernel\w+):", asm, flags=re.M)))
    assert len(names) == 24, names   # {bf16, fp16, fp32} x align_corners x strict order x range check
    for name in names:
        a = asm.index(name + ":")
        body = asm[a:asm.index(".Lfunc_end", a)]
        assert "scratch_" not in body, f"{name}: scratch (spill) operations in the kernel"
        meta = asm[asm.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size\s+(\d+)", meta).group(1))
        # bf16 / fp16: 1024 threads, two workgroups per CU = 8 waves per SIMD = 64 registers; fp32: 512 threads (2 pixels per thread, round 4) = 4 waves per
        # SIMD = 128 registers (the unified file: `next_free_vgpr` counts the accumulation registers the allocator parks values in)
        fp32 = "render_band_kernelIf" in name
        assert vgpr <= (128 if fp32 else 64), (name, vgpr)   # (a private segment may be RESERVED -- a frame object whose accesses were optimised away -- but not used)
        assert 2 * lds <= 160 * 1024, (name, lds)
        # the plane loop: one s_barrier per plane step, the DMA and the taps inside it
        assert body.count("s_barrier") >= 2 and "buffer_load_dwordx4" in body and " lds" in body
    # the software-pipelined taps (round 4): no instruction may touch the destination of an LDS read that is still in flight
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_pipe", os.path.join(ROOT, "tools", "isa_pipe.py"))
    isa_pipe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(isa_pipe)
    piped = 0
    for name in names:
        a = asm.index(name + ":")
        body = asm[a:asm.index(".Lfunc_end", a)].split("\n")
        assert isa_pipe.check(body, name) == [], name
        piped += any(re.search(r"s_waitcnt lgkmcnt\([1-9]\d*\)", l) for l in body)
    assert piped == 12, piped   # the default-mode instances (storage type x align_corners x range check) are the pipelined ones
    shutil.rmtree(tmp_path, ignore_errors=True)
