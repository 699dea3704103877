"""Build variants of the lane kernel (extra -D flags / -mllvm options) into tools/.prof/var_<name>.so for side-by-side
timing on the GPU box:  python tools/lane_variants.py name1="-DX=1 -DY=0" name2="..."   (objects of the other two
translation units are reused from csrc/build/).  Then on the GPU:  for v in tools/.prof/var_*.so; QMPC_LIB=$v ..."""
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "quaternion-mpc_amd" / "csrc"
OUT = REPO / "tools" / ".prof"
OUT.mkdir(parents=True, exist_ok=True)
procs = []
for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    obj = CSRC / "build" / f"lane_var_{name}.o"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-mllvm", "-disable-lsr", *flags.split(),
           "-c", str(CSRC / "qmpc_lane.hip"), "-o", str(obj)]
    procs.append((name, obj, subprocess.Popen(cmd)))
for name, obj, p in procs:
    assert p.wait() == 0, name
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", str(OUT / f"var_{name}.so"),
                    str(CSRC / "build" / "qmpc_hip.o"), str(CSRC / "build" / "qmpc_loop_fused.o"), str(obj)], check=True)
    print("built", name)
