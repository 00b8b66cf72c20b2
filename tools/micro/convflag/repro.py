#!/usr/bin/env python3
"""Demonstrates the cause of the round-1 maskconv_cl staging fault on the GPU box.

The interim build (commit 346739d, rebuilt by build.sh into build/libconv_flagbuild.so) chose the feature-window mode with a
RUNTIME flag inside the staging loop.  hipcc (ROCm 7.2) evaluated that wave-uniform flag with a VALU compare into an SGPR
lane mask inside the FIRST output row's staging loop -- under that loop's EXEC, which in a wave's last iteration holds only
the lanes with i = tid + 256 k < KG * PW -- hoisted it, and re-used the mask in the SECOND row's loop as
`s_and_b64 vcc, exec, mask; s_cbranch_vccnz <global-load path>` (build/conv_cl_flagbuild.s, .LBB4_13 / .LBB4_20).  When the lanes
that were active in that last iteration (tid < m = KG * PW mod 256, m < 64) are all padding frames in the second row's first
iteration (m <= KG * pad_left), vcc is 0, the wave takes the buffer-load (window) path whose descriptor has num_records = 0 in
multi-channel mode, and stages ZEROS for its 64 granules.  Kernel 3 / stride 2 / SAME at Cin = 16 has m = 2, pad_left = 1.

Usage (GPU box):  python tools/micro/convflag/repro.py
Prints, per library, the max error against a float64 convolution and which outputs were wrong."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from myrtlespeech_amd import _lib  # noqa: E402
from myrtlespeech_amd.model.cnn import pad_same  # noqa: E402

P = ctypes.c_void_p


def bind(path):
    lib = ctypes.CDLL(path)
    lib.ms_maskconv_cl_packed_bytes.restype = ctypes.c_size_t
    lib.ms_maskconv_cl_packed_bytes.argtypes = [ctypes.c_int] * 4
    lib.ms_maskconv_cl_pack.argtypes = [P, P] + [ctypes.c_int] * 4 + [P]
    lib.ms_maskconv_cl_workspace_bytes.restype = ctypes.c_size_t
    lib.ms_maskconv_cl_workspace_bytes.argtypes = [ctypes.c_int] * 4
    lib.ms_maskconv_cl_forward.argtypes = [P] * 5 + [ctypes.c_int] * 16 + [ctypes.c_float, ctypes.c_float, P, ctypes.c_size_t, P]
    lib.ms_last_error.restype = ctypes.c_char_p
    return lib


def run(lib, x, w, b, lens, st, pt, pf):
    n, cin, fin, tin = x.shape
    cout, _, kf, kt = w.shape
    fout = fin + sum(pf) - kf + 1
    tout = (tin + sum(pt) - kt) // st + 1
    pk = torch.empty(lib.ms_maskconv_cl_packed_bytes(cout, cin, kf, kt), dtype=torch.uint8, device="cuda")
    assert lib.ms_maskconv_cl_pack(w.data_ptr(), pk.data_ptr(), cout, cin, kf, kt, None) == 0
    ws = torch.empty(lib.ms_maskconv_cl_workspace_bytes(n, cin, fin, tin), dtype=torch.uint8, device="cuda")
    y = torch.full((n, cout, fout, tout), float("nan"), device="cuda")
    rc = lib.ms_maskconv_cl_forward(x.data_ptr(), lens.data_ptr(), pk.data_ptr(), b.data_ptr(), y.data_ptr(), n, cin, fin, tin,
                                    cout, fout, tout, kf, kt, 1, st, 1, 1, pf[0], pt[0], 0, 0.0, 0.0, ws.data_ptr(), ws.numel(), None)
    assert rc == 0, lib.ms_last_error()
    torch.cuda.synchronize()
    return y


def main():
    torch.manual_seed(0)
    for cin, kt, st in ((16, 3, 2), (32, 5, 3), (16, 3, 1)):     # the last one has m = 4 > KG * pad_left = 2: no fault expected
        cout, kf, fin, tin = 32, 3, 6, 300
        pt, pf = pad_same(tin, kt, st, 1), pad_same(fin, kf, 1, 1)
        x = torch.randn(1, cin, fin, tin, device="cuda") + 3.0   # offset: a zeroed input granule shows in every output it feeds
        w = torch.randn(cout, cin, kf, kt, device="cuda") * 0.1
        b = torch.zeros(cout, device="cuda")
        lens = torch.tensor([tin], dtype=torch.int32, device="cuda")
        xp = torch.nn.functional.pad(x.double().cpu(), (pt[0], pt[1], pf[0], pf[1]))
        want = torch.nn.functional.conv2d(xp, w.double().cpu(), None, stride=(1, st))
        kg, pw = cin // 8, 127 * st + (kt - 1) + 1
        print(f"Cin {cin} KT {kt} stride {st} SAME pad_t {pt}: KG*PW = {kg * pw}, m = KG*PW mod 256 = {kg * pw % 256}, "
              f"KG*pad_left = {kg * pt[0]}")
        for name, path in (("current build", _lib.LIB_PATH),
                           ("interim flag build (346739d)", os.path.join(HERE, "build", "libconv_flagbuild.so"))):
            y = run(bind(path), x, w, b, lens, st, pt, pf).double().cpu()
            err = (y - want).abs()
            bad = (err > 1e-2).nonzero()
            print(f"   {name}: max |err| {float(err.max()):.3e}; wrong outputs {bad.shape[0]}")
            if bad.shape[0]:
                rows = sorted(set(int(v) for v in bad[:, 2]))
                frames = sorted(set(int(v) for v in bad[:, 3]))
                print(f"      wrong output feature rows {rows}; wrong output frames {frames[0]}..{frames[-1]} ({len(frames)} frames): "
                      f"wave 0 staged zeros for its first 64 granules = {64 // kg} input frames of the workgroup's second row")


if __name__ == "__main__":
    main()
