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
print("ISA metadata of the final build (hipcc --offload-arch=gfx950 -O2 <per-unit flags of __graft_entry__.py> -S --cuda-device-only; .amdgpu_metadata notes), one row per kernel")
print("columns: kernel | vgpr_count | agpr_count | sgpr_count | vgpr_spill_count | sgpr_spill_count | private_segment_fixed_size (scratch bytes) | group_segment_fixed_size (static LDS)")
WAVE = ["-mllvm", "-disable-machine-licm", "-mllvm", "-disable-machine-sink"]
for tu, extra in (("qmpc_hip.hip", WAVE), ("qmpc_loop_fused.hip", WAVE), ("qmpc_wform.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
                  ("qmpc_lane.hip", ["-mllvm", "-disable-lsr", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DQL_UNIT=1"]),
                  ("qmpc_lane_ref.hip", ["-mllvm", "-disable-lsr", "-DQL_UNIT=2"])):
    with tempfile.TemporaryDirectory() as d:
        asm = Path(d) / "tu.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", *extra, "-std=c++17", "-S", "--cuda-device-only", "-o", str(asm),
                        str(CSRC / tu)], check=True, stderr=subprocess.DEVNULL)
        txt = asm.read_text()
    print(f"---- translation unit {tu} (extra flags: {' '.join(extra)})" + (" (the kernel calls its passes as functions: the per-function figures follow the kernel rows)" if "lane" in tu else ""))
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
        blk = m.group(0)
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        name = subprocess.run(["/usr/bin/c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{name} | {g('vgpr_count')} | {g('agpr_count')} | {g('sgpr_count')} | {g('vgpr_spill_count')} | {g('sgpr_spill_count')} | "
              f"{g('private_segment_fixed_size')} | {g('group_segment_fixed_size')}")
    if "lane" in tu:      # non-inlined device functions: registers and scratch from the .set directives of the listing
        for fn in re.findall(r"^\t\.set (\.L_ZN4qmpc4lane\w+)\.num_vgpr, (\d+)", txt, re.M):
            sym = fn[0]
            get = lambda k: re.search(re.escape(sym) + r"\." + k + r", (\d+)", txt).group(1)
            name = subprocess.run(["/usr/bin/c++filt", sym[2:]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "").replace("bool ", "")
            print(f"  function {name} | vgpr {get('num_vgpr')} | agpr {get('num_agpr')} | sgpr {get('numbered_sgpr')} | scratch {get('private_seg_size')} B")
