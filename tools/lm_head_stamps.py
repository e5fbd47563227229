"""Where a workgroup of the forward LM-head kernel spends its time: a diagnostic build of the library
(-DPRL_GEMM_STAMPS) reads the 100 MHz clock per work item at the top of the item (its first loads are in flight
already), when its first K-tile has landed, after the K loop, and when its stores (and the next item's first
loads) are issued, and sums the four phases per workgroup.

    PRL_HIP_LIB=prime-rl-cse_amd/lib/variants/libprl_stamps.so python tools/lm_head_stamps.py [T] [H] [V]
"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import _capi, lm_head as L  # noqa: E402

t, h, v = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 1024, 151936)
dev = torch.device("cuda:0")
lib = ctypes.CDLL(str(_capi.os.environ["PRL_HIP_LIB"]))
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((t, h), generator=gen, device=dev).to(torch.bfloat16)
w = (torch.randn((v, h), generator=gen, device=dev) * 0.05).to(torch.bfloat16)
stamps = torch.zeros(256 * 8, dtype=torch.int64, device=dev)  # per workgroup: 4 phase sums, items, first, last
for _ in range(2):
    L.lm_head_fwd(x, w)
lib.prl_lm_head_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
L.lm_head_fwd(x, w)
torch.cuda.synchronize()
lib.prl_lm_head_set_stamps(None)
s = stamps.view(256, 8).cpu().double()
items = s[:, 4]
span = (s[:, 6].max() - s[:, 5].min()) * 0.01
print(f"{int(items.sum())} work items on {int((items > 0).sum())} workgroups; span {float(span):.1f} us; workgroups start "
      f"within {float((s[:, 5].max() - s[:, 5].min()) * 0.01):.2f} us and end within "
      f"{float((s[:, 6].max() - s[:, 6].min()) * 0.01):.2f} us of each other")
names = ("stores issued -> top of the next item", "top -> first K-tile landed", "K loop", "write-back + next loads issued")
for i, name in enumerate(names):
    per_item = s[:, i] * 0.01 / items
    print(f"{name:40s} per item: mean {float(per_item.mean()):6.2f} us  min {float(per_item.min()):6.2f}  max {float(per_item.max()):6.2f}")
print(f"whole item: {float((s[:, :4].sum(1) * 0.01 / items).mean()):.2f} us")
mhz = s[:, 7] / ((s[:, 6] - s[:, 5]) * 0.01)
print(f"shader clock while the kernel ran: mean {float(mhz.mean()):.0f} MHz  min {float(mhz.min()):.0f}  max {float(mhz.max()):.0f}")
# shader cycles per interval of a K-tile (wave 0 of every workgroup; an interval lasts until BOTH groups of waves have
# reached the next barrier, so it is the longer of the two sections named)
if hasattr(lib, "prl_lm_head_set_section_stamps"):
    sec = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
    lib.prl_lm_head_set_section_stamps(ctypes.c_void_p(sec.data_ptr()))
    lib.prl_lm_head_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    L.lm_head_fwd(x, w)
    torch.cuda.synchronize()
    lib.prl_lm_head_set_section_stamps(None)
    lib.prl_lm_head_set_stamps(None)
    c = sec.view(256, 8).cpu().double()
    tiles = float(items.mean()) * (h // 64)
    names = {1: "load 0 | MFMA 3", 2: "MFMA 0 | load 0", 3: "load 1 | MFMA 0", 4: "MFMA 1 | load 1", 5: "load 2 | MFMA 1",
             6: "MFMA 2 | load 2", 7: "load 3 | MFMA 2", 0: "MFMA 3 | load 3"}
    print("cycles per interval of a K-tile (waves 0-3 | waves 4-7), with the stamps' own cost in every one:")
    for k in (1, 2, 3, 4, 5, 6, 7, 0):
        print(f"  {names[k]:18s} {float(c[:, k].mean()) / tiles:7.1f}")
    print(f"  sum {float(c.sum(1).mean()) / tiles:7.1f} per K-tile; 16 MFMAs are 256 cycles")
