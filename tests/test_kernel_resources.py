"""Guard the occupancy of the bandwidth-bound kernels: they run as ONE round of workgroups (1954 chunks on 256 CUs x 8 slots at
N = 4 M), which needs <= 64 VGPRs per thread.  A run-time branch added to the direction kernel once pushed it to 74 VGPRs (6 workgroups
per CU: 22.6 -> 25.4 us, 16.8 -> 16.2 k CG it/s) without failing a single parity test; this compiles the file for gfx950 and reads
the compiler's own resource report (no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _vgprs(src):
    """{demangled kernel name up to '(': VGPRs} of one translation unit, from hipcc's -Rpass-analysis=kernel-resource-usage."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
               "-Wno-unused-function", "-fno-gpu-rdc", "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "x.o"),
               "-Rpass-analysis=kernel-resource-usage"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    usage, sgprs, scratch, name = {}, {}, {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"TotalSGPRs: (\d+)", line)
        if m and name:
            sgprs[name] = int(m.group(1))
        m = re.search(r"\bVGPRs: (\d+)", line)
        if m and name:
            usage[name] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    filt = shutil.which("c++filt")
    names = subprocess.run([filt], input="\n".join(usage), capture_output=True, text=True).stdout.splitlines()
    out = {d.split("(")[0]: v for d, v in zip(names, usage.values())}
    _vgprs.sgprs = {d.split("(")[0]: sgprs.get(k, 0) for d, k in zip(names, usage)}
    _vgprs.scratch = {d.split("(")[0]: scratch.get(k, 0) for d, k in zip(names, usage)}
    return out


HOT = {
    "hipk_cg.hip": ["void hipk_cg_update_kernel<double, false, false>", "void hipk_cg_update_kernel<double, false, true>",
                    "void hipk_cg_direction_kernel<double, false, false, false>",
                    "void hipk_cg_direction_kernel<double, false, true, false>",
                    "hipk_cg_update_fx_kernel", "hipk_cg_direction_fx_kernel"],
    "hipk_bicgstab.hip": ["void hipk_bi_supdate_kernel<double, false, false>", "void hipk_bi_xupdate_kernel<double, false, false>"],
    "hipk_gmres.hip": ["void hipk_gm_multidot_stream_kernel<double, 8>", "void hipk_gm_update_stream_kernel<double, 8>",
                       "void hipk_gm_normalize_kernel<double>"],
    "hipk_api.hip": ["void hipk_spmv_kernel<double, 1280, true>", "void hipk_spmv_sell_pair_kernel<double, 5, true, 1>",
                     "void hipk_spmv_sell_pair_kernel<double, 5, true, 2>", "void hipk_spmv_sell_pair_kernel<double, 5, true, -1>",
                     "void hipk_spmv_sell_wide_kernel<5, 1, 0>", "void hipk_spmv_sell_wide_kernel<5, 2, 0>",
                     "void hipk_spmv_sell_wide_kernel<5, -1, 0>", "void hipk_spmv_sell_wide_kernel<5, 1, 1>",
                     "void hipk_spmv_sell_wide_kernel<5, 2, 1>", "void hipk_spmv_sell_wide_kernel<5, -1, 1>",
                     "void hipk_spmv_sell_wide_kernel<8, 1, 0>", "void hipk_spmv_sell_wide_kernel<8, 2, 0>"],
}


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="hipcc / c++filt not installed")
@pytest.mark.parametrize("src", sorted(HOT))
def test_bandwidth_bound_kernels_keep_eight_workgroups_per_cu(src):
    got = _vgprs(src)
    for k in HOT[src]:
        assert k in got, (k, sorted(got)[:60])
        assert got[k] <= 64, f"{k}: {got[k]} VGPRs (> 64: fewer than 8 workgroups per CU, the chunks no longer run as one round)"
        # MI355X admits only SEVEN 256-thread workgroups per CU at 82-96 scalar registers although the occupancy API says 8
        # (MI355X_MICROARCH.md "Residency"; round 3 found the GMRES multi-dot at 96 and the fused-exchange CG kernels at 83-84)
        assert _vgprs.sgprs[k] <= 80, f"{k}: {_vgprs.sgprs[k]} SGPRs (> 80: the hardware admits fewer than 8 workgroups per CU)"


# The one-launch loops of mid-size systems (hipk_cg_mid.h, hipk_bi_mid.h) run 1024 threads per workgroup, one workgroup per CU:
# four wavefronts per SIMD = 128 VGPRs.  The instantiations the stencil matrices take must fit them without (or nearly without)
# scratch: their phases are VALU-issue bound, every spilled register is a memory round trip per iteration.
MID = {
    "hipk_cg.hip": {"void hipk_cg_mid_kernel<double, 5, 1, false>": 0, "void hipk_cg_mid_kernel<double, 7, 1, false>": 0,
                    "void hipk_cg_mid_kernel<double, 9, 1, false>": 0, "void hipk_cg_mid_kernel<double, 5, 2, false>": 48,
                    "void hipk_cg_mid_kernel<double, 5, 1, true>": 0, "void hipk_cg_mid_kernel<double, 7, 1, true>": 8, "void hipk_cg_mid_kernel<float, 5, 1, false>": 0,
                    "void hipk_cg_mid_kernel<float, 12, 1, false>": 0},
    "hipk_bicgstab.hip": {"void hipk_bi_mid_kernel<double, 5, false>": 8, "void hipk_bi_mid_kernel<double, 7, false>": 48, "void hipk_bi_mid_kernel<double, 5, true>": 8,
                          "void hipk_bi_mid_kernel<float, 5, false>": 0},
    "hipk_gmres.hip": {"void hipk_gm_mid_kernel<double, 5, false>": 0, "void hipk_gm_mid_kernel<double, 7, false>": 32, "void hipk_gm_mid_kernel<double, 5, true>": 16,
                       "void hipk_gm_mid_kernel<float, 5, false>": 0},
}


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="hipcc / c++filt not installed")
@pytest.mark.parametrize("src", sorted(MID))
def test_one_launch_loops_fit_four_wavefronts_per_simd(src):
    got = _vgprs(src)
    for k, max_scratch in MID[src].items():
        assert k in got, (k, sorted(got)[:60])
        assert got[k] <= 128, f"{k}: {got[k]} VGPRs (> 128: a 1024-thread workgroup no longer fits a CU)"
        assert _vgprs.scratch[k] <= max_scratch, f"{k}: {_vgprs.scratch[k]} bytes of scratch per lane (budget {max_scratch})"
