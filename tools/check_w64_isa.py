"""Compile winograd64_conv.hip to ISA and count accumulator moves per kernel: the k-loops must have none (256 reads and
256 (+2) writes per kernel are the epilogue's and the prologue's). hipcc's allocation of this kernel flips with small edits."""
import re, subprocess, sys, tempfile
from pathlib import Path

src = Path(__file__).resolve().parent.parent / "gance_amd" / "csrc" / "winograd64_conv.hip"
with tempfile.NamedTemporaryFile(suffix=".s") as out:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S", str(src), "-o", out.name],
                   check=True, cwd=src.parent, stderr=subprocess.DEVNULL)
    lines = Path(out.name).read_text().split("\n")
bad = False
for i, line in enumerate(lines):
    m = re.match(r"^(_ZN5gance\w+):", line)
    if not m:
        continue
    end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
    body = lines[i:end]
    reads, writes, moves = (sum(op in l for l in body) for op in ("v_accvgpr_read", "v_accvgpr_write", "v_accvgpr_mov"))
    spills = next((l.split(":")[1].strip() for l in lines[end:] if "vgpr_spill_count" in l and "sgpr" not in l), "?")
    print(f"{m.group(1)[9:45]:40s} accvgpr read {reads:4d} write {writes:4d} mov {moves:3d}")
    if "winograd64" in m.group(1) and "coef" not in m.group(1) and (reads > 260 or writes > 260 or moves):
        bad = True
sys.exit(1 if bad else 0)
