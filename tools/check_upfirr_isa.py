"""Compile upfir_split_roles.hip to ISA (no GPU needed) and check what the role-split experiment depends on: eight waves per block = two
per SIMD (<= 256 registers per lane for the matrix waves' 64 accumulators + 108 weight-fragment registers + two sets of patch fragments),
no scratch at all, the K loop's MFMAs all there (2 chunk parities x (216 products + the halo tile's 54))."""
import re, subprocess, sys, tempfile
from pathlib import Path

src = Path(__file__).resolve().parent.parent / "gance_amd" / "csrc" / "upfir_split_roles.hip"
with tempfile.NamedTemporaryFile(suffix=".s") as out:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S", str(src), "-o", out.name],
                   check=True, cwd=src.parent, stderr=subprocess.DEVNULL)
    text = Path(out.name).read_text()
bad = False
for name in ("upfirr_fused_kernel", "upfirr_fused_noise_kernel"):
    start = text.index(f"_ZN5gance{len(name)}{name}ENS_9UpFirArgsE:")
    end = text.index(".Lfunc_end", start)
    meta = text[end:end + 6000]
    vgprs = int(re.search(r"; NumVgprs: (\d+)", meta).group(1))
    agprs = int(re.search(r"; NumAgprs: (\d+)", meta).group(1))
    scratch = int(re.search(r"; ScratchSize: (\d+)", meta).group(1))
    occupancy = int(re.search(r"; Occupancy: (\d+)", meta).group(1))
    mfmas = len(re.findall(r"^\s+v_mfma_f32_16x16x32_bf16", text[start:end], re.M))
    print(f"{name:28s} VGPRs {vgprs:3d} AGPRs {agprs:3d} scratch {scratch:4d} B occupancy {occupancy} MFMAs {mfmas}")
    if vgprs + agprs > 256 or scratch != 0 or occupancy < 2 or mfmas != 540:
        bad = True
sys.exit(1 if bad else 0)
