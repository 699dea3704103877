#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every gfx950 kernel of the library, from the compiler's own metadata.
Runs in the build container (no GPU):  python tools/isa_metadata.py > profiles/rNN_isa_metadata.txt"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / "quaternion-mpc_amd" / "csrc"
print("ISA metadata of the final build (hipcc --offload-arch=gfx950 -O2 -mllvm -disable-machine-licm -mllvm -disable-machine-sink -S --cuda-device-only; .amdgpu_metadata notes), one row per kernel")
print("columns: kernel | vgpr_count | agpr_count | sgpr_count | vgpr_spill_count | sgpr_spill_count | private_segment_fixed_size (scratch bytes) | group_segment_fixed_size (static LDS)")
for tu in ("qmpc_hip.hip", "qmpc_loop_fused.hip"):
    with tempfile.TemporaryDirectory() as d:
        asm = Path(d) / "tu.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-mllvm", "-disable-machine-licm", "-mllvm", "-disable-machine-sink", "-std=c++17", "-S", "--cuda-device-only", "-o", str(asm),
                        str(CSRC / tu)], check=True, stderr=subprocess.DEVNULL)
        txt = asm.read_text()
    print(f"---- translation unit {tu}")
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
        blk = m.group(0)
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        name = subprocess.run(["/usr/bin/c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{name} | {g('vgpr_count')} | {g('agpr_count')} | {g('sgpr_count')} | {g('vgpr_spill_count')} | {g('sgpr_spill_count')} | "
              f"{g('private_segment_fixed_size')} | {g('group_segment_fixed_size')}")
