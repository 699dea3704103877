"""Build variants of the lane kernel (extra -D flags / -mllvm options) into tools/.prof/var_<name>.so for side-by-side
timing on the GPU box:  python tools/lane_variants.py name1="-DX=1 -DY=0" name2="..."   (objects of the other three
translation units are reused from csrc/build/).  Then on the GPU:  for v in tools/.prof/var_*.so; QMPC_LIB=$v ..."""
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "quaternion-mpc_amd" / "csrc"
OUT = REPO / "tools" / ".prof"
OUT.mkdir(parents=True, exist_ok=True)
procs = []
import tempfile
for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    src_dir = CSRC
    if "@" in name:      # name@rev: the lane sources as of that git revision
        name, rev = name.split("@")
        tmp = Path(tempfile.mkdtemp(prefix="lane_" + name))
        (tmp / "include").mkdir()
        src_dir = tmp / "quaternion-mpc_amd" / "csrc"
        src_dir.mkdir(parents=True)
        for f in ("qmpc_lane.hip", "qmpc_lane_core.h", "qmpc_params_dev.h"):
            (src_dir / f).write_bytes(subprocess.check_output(["git", "-C", str(REPO), "show", f"{rev}:quaternion-mpc_amd/csrc/{f}"]))
        (tmp / "include" / "qmpc.h").write_bytes((REPO / "include" / "qmpc.h").read_bytes())
    obj = CSRC / "build" / f"lane_var_{name}.o"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-mllvm", "-disable-lsr", *flags.split(),
           "-c", str(src_dir / "qmpc_lane.hip"), "-o", str(obj)]
    procs.append((name, obj, subprocess.Popen(cmd)))
for name, obj, p in procs:
    assert p.wait() == 0, name
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", str(OUT / f"var_{name}.so"),
                    str(CSRC / "build" / "qmpc_hip.o"), str(CSRC / "build" / "qmpc_loop_fused.o"),
                    str(CSRC / "build" / "qmpc_wform.o"), str(obj)], check=True)
    print("built", name)
