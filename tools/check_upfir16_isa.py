"""Compile upfir16_fused.hip to ISA (no GPU needed) and check what its speed depends on: two blocks per CU (<= 256 registers
per lane, nothing in AGPR-only form that would not fit), no scratch access in the K loop or the epilogue passes (a spill reload there is a
vector-memory load: its wait drains the LDS-DMA ring; a first version reloaded seventeen values per chunk), the K loop's MFMAs all there."""
import re, subprocess, sys, tempfile
from pathlib import Path

src = Path(__file__).resolve().parent.parent / "gance_amd" / "csrc" / "upfir16_fused.hip"
with tempfile.NamedTemporaryFile(suffix=".s") as out:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S", str(src), "-o", out.name],
                   check=True, cwd=src.parent, stderr=subprocess.DEVNULL)
    text = Path(out.name).read_text()
bad = False
names = [f"upfir16_fused{geo}{pre}{noise}_kernel" for geo in ("", "_w32", "_w16") for pre in ("", "_pre") for noise in ("", "_noise")] + [f"upfir16x_fused{geo}_pre{noise}_kernel" for geo in ("", "_w32") for noise in ("", "_noise")]
for name in names:
    start = text.index(f"_ZN5gance{len(name)}{name}ENS_9UpFirArgsE:")
    end = text.index(".Lfunc_end", start)
    meta = text[end:end + 6000]
    vgprs = int(re.search(r"; NumVgprs: (\d+)", meta).group(1))
    agprs = int(re.search(r"; NumAgprs: (\d+)", meta).group(1))
    scratch = int(re.search(r"; ScratchSize: (\d+)", meta).group(1))
    occupancy = int(re.search(r"; Occupancy: (\d+)", meta).group(1))
    mfmas = len(re.findall(r"^\s+v_mfma_f32_16x16x4_f32", text[start:end], re.M))
    # scratch traffic where it hurts: in a basic block that holds MFMAs or a barrier (the K loop, the epilogue passes)
    hot, block = 0, []
    for line in text[start:end].split("\n") + [".LBB_end:"]:
        if re.match(r"^\.LBB\w+:", line):
            if any(i.startswith(("v_mfma", "s_barrier")) for i in block):
                hot += sum(i.startswith("scratch_") for i in block)
            block = []
        elif re.match(r"^\s+[a-z]", line):
            block.append(line.split()[0])
    print(f"{name:36s} VGPRs {vgprs:3d} AGPRs {agprs:3d} scratch {scratch:4d} B ({hot} accesses beside MFMAs / barriers) occupancy {occupancy} MFMAs {mfmas}")
    if vgprs + agprs > 256 or scratch > 64 or hot != 0 or occupancy < 2 or mfmas < 144:
        bad = True
sys.exit(1 if bad else 0)
