"""Worker of tests/test_gpu_parity.py::test_two_process_rccl_gather_on_one_gpu: rank `r` of an `n`-rank RCCL
communicator created straight from librccl (ncclCommInitRank), every rank on GPU 0.  Each rank solves its own shard
through the C ABI and calls qmpc_gather (ncclAllGather).  Exit code 0 = gathered data verified; 3 = RCCL refused to
build a communicator with several ranks on one device (prints the ncclResult); anything else = failure."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
from conftest import load_pkg  # noqa: E402

rank, world, uid_path, counts = int(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3]), [int(c) for c in sys.argv[4].split(",")]
import torch  # noqa: E402

pkg = load_pkg()
lib = pkg.load_library()
rccl = None
for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
    try:
        rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
        break
    except OSError:
        continue
if rccl is None:
    sys.exit(4)


class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


u = Uid()
if rank == 0:
    assert rccl.ncclGetUniqueId(C.byref(u)) == 0
    tmp = uid_path.with_suffix(".tmp")
    tmp.write_bytes(bytes(u))
    tmp.rename(uid_path)
else:
    t0 = time.time()
    while not uid_path.exists():
        if time.time() - t0 > 60:
            sys.exit(5)
        time.sleep(0.05)
    C.memmove(C.byref(u), uid_path.read_bytes(), 128)
torch.cuda.set_device(0)
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
rccl.ncclGetErrorString.restype = C.c_char_p
comm = C.c_void_p()
rc = rccl.ncclCommInitRank(C.byref(comm), world, u, rank)
if rc != 0:
    print(f"rank {rank}: ncclCommInitRank({world} ranks on one device) -> {rc}: {rccl.ncclGetErrorString(rc).decode()}", flush=True)
    sys.exit(3)
# ragged shards: every rank pads its block to the largest shard (the collective moves equal counts)
first = sum(counts[:rank])
n, m = counts[rank], max(counts)
p = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
s = pkg.Solver(p, m, device=0, lib=lib)
rec = pkg.random_go1_trot_states(n, config_id=2, first=first)
d_in = torch.from_numpy(rec.view(np.uint8).reshape(n, -1).copy()).cuda()
IW = pkg.INFO_DTYPE.itemsize // 8
blk = torch.zeros(m * (12 + IW), dtype=torch.float64, device="cuda")
st = torch.cuda.Stream()
s.solve_device(n, d_in.data_ptr(), blk[:m * 12].data_ptr(), blk[m * 12:].data_ptr(), st.cuda_stream)
d_all = torch.zeros(world * blk.numel(), dtype=torch.float64, device="cuda")
s.gather(comm.value, blk.data_ptr(), blk.numel(), d_all.data_ptr(), st.cuda_stream)   # stream-ordered after the solve
st.synchronize()
allb = d_all.view(world, -1).cpu().numpy()
# reference: the whole batch on this one GPU through the host-buffer call
total = sum(counts)
sf = pkg.Solver(p, total, device=0, lib=lib)
f_full, i_full = sf.solve(pkg.random_go1_trot_states(total, config_id=2))
ok = True
for r in range(world):
    fr = allb[r, :m * 12].reshape(m, 12)[:counts[r]]
    ir = np.ascontiguousarray(allb[r, m * 12:]).view(pkg.INFO_DTYPE)[:counts[r]]
    o = sum(counts[:r])
    ok = ok and np.array_equal(fr, f_full[o:o + counts[r]]) and np.array_equal(ir["iterations"], i_full["iterations"][o:o + counts[r]])
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
s.close(); sf.close()
print(f"rank {rank}: gathered {world} blocks, ok={ok}", flush=True)
sys.exit(0 if ok else 1)
