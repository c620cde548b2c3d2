"""
The Winograd kernel's register allocation is a knife edge (256 accumulators + 256 other registers): small edits have
made hipcc stage accumulator tiles through VGPRs, 80 ... 120 copies per loop trip, +13 ... +20 % run time, with every
parity test still green. This check cross-compiles the kernel (no GPU needed) and counts accumulator moves per kernel:
only the epilogue's 256 reads and the clears may be there (tools/check_w64_isa.py).
"""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

REPO_ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="needs hipcc")
def test_winograd_kernels_keep_their_accumulators_in_agprs() -> None:
    result = subprocess.run([sys.executable, str(REPO_ROOT / "tools" / "check_w64_isa.py")], capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "winograd64_rgb_kernel" in result.stdout and "winograd64_c32_rgb_kernel" in result.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="needs hipcc")
def test_winograd43_kernels_do_not_spill_and_run_two_waves_per_simd() -> None:
    """
    The F(4x4,3x3) kernel (144 accumulators + 72 operand registers per wave, two waves per SIMD) is correct with spills but
    slow: a scratch reload in the k-steps is a vector-memory load whose wait drains the LDS-DMA ring (an epilogue under a
    conditional inside the stream loop cost 244 spilled registers). tools/check_w43_isa.py cross-compiles and checks.
    """
    result = subprocess.run([sys.executable, str(REPO_ROOT / "tools" / "check_w43_isa.py")], capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "winograd43_rgb_kernel" in result.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="needs hipcc")
def test_upfir16_kernels_fit_two_blocks_per_cu_without_scratch() -> None:
    """
    The 16-channel fused up kernel (upfir16_fused.hip) is built for two blocks per CU: 128 accumulators + two sets of operand
    fragments in at most 256 registers, and no scratch (its first version reloaded seventeen spilled values at the top of
    every chunk of its K loop). tools/check_upfir16_isa.py cross-compiles and checks.
    """
    result = subprocess.run([sys.executable, str(REPO_ROOT / "tools" / "check_upfir16_isa.py")], capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "upfir16_fused_pre_kernel" in result.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="needs hipcc")
def test_role_split_up_kernel_fits_two_waves_per_simd_without_scratch() -> None:
    """
    The role-split experiment (upfir_split_roles.hip) runs eight waves per block, two per SIMD: its matrix waves hold 64 accumulators, 108
    weight-fragment registers and two sets of patch fragments in at most 256 registers; a spill there would be a vector-memory access
    inside the row loop. tools/check_upfirr_isa.py cross-compiles and checks registers, scratch and the K loop's MFMA count.
    """
    result = subprocess.run([sys.executable, str(REPO_ROOT / "tools" / "check_upfirr_isa.py")], capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "upfirr_fused_kernel" in result.stdout

