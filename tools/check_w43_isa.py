"""Compile winograd43_conv.hip to ISA (no GPU needed) and check what its speed depends on: no scratch (a spill reload is a
vector-memory load: the wait behind it drains the LDS-DMA ring), exactly two waves per SIMD (171 ... 256 registers: the kernel
derives a wave's work from its SIMD and its ticket there), no 64-bit register
moves in the blocks that hold MFMAs (accumulator copies through phi nodes)."""
import re, subprocess, sys, tempfile
from pathlib import Path

src = Path(__file__).resolve().parent.parent / "gance_amd" / "csrc" / "winograd43_conv.hip"
with tempfile.NamedTemporaryFile(suffix=".s") as out:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S", str(src), "-o", out.name],
                   check=True, cwd=src.parent, stderr=subprocess.DEVNULL)
    text = Path(out.name).read_text()
bad = False
for name in ("winograd43_kernel", "winograd43_rgb_kernel", "winograd43_w32_kernel", "winograd43_w32_rgb_kernel"):
    start = text.index(f"_ZN5gance{len(name)}{name}ENS_8ConvArgsE:")
    end = text.index(".Lfunc_end", start)
    body = text[start:end].split("\n")
    meta = text[end:end + 6000]
    vgprs = int(re.search(r"; NumVgprs: (\d+)", meta).group(1))
    scratch = int(re.search(r"; ScratchSize: (\d+)", meta).group(1))
    blocks, current = [], []
    for line in body:
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append(current)
            current = []
        elif re.match(r"^\s+[a-z]", line):
            current.append(line.split()[0])
    blocks.append(current)
    mfma_blocks = [b for b in blocks if sum(i.startswith("v_mfma") for i in b) >= 4]
    moves = sum(sum(i.startswith("v_mov_b64") for i in b) for b in mfma_blocks)
    mfmas = sum(sum(i.startswith("v_mfma") for i in b) for b in blocks)
    print(f"{name:24s} VGPRs {vgprs:3d} scratch {scratch:4d} B, MFMAs {mfmas} (36 per k-step copy), 64-bit moves beside MFMAs: {moves}")
    # (> 170 registers: a third wave must not fit on a SIMD -- the kernel takes a wave's tile row from the SIMD it runs on)
    if vgprs > 256 or vgprs <= 170 or scratch != 0 or mfmas < 72 or moves > 8:
        bad = True
sys.exit(1 if bad else 0)
