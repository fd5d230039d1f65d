#!/usr/bin/env python
"""Review r5 item 4, the measurement: can the glue passes of the fused ResNet path (csrc/glue.hip: bias + ReLU, bias + shortcut +
ReLU, threshold, junction add + threshold) disappear into the epilogues of composable_kernel convolutions?

For the bottleneck convolutions of ResNet-50 at batch N (NHWC, fp32) this times, in ONE process on the same tensors,

    unfused   what backbones/fused.py runs today: MIOpen's convolution (find mode) + the glue kernel of libta_hip.so
    fused     every tile configuration of CK's own f32 instance lists, instantiated with the epilogue as CDE operation
              (tools/ck_probe: y = clamp_min((acc + b) [+ shortcut], 0) forward; dx = act <= 0 ? 0 : acc [+ other] backward --
              the activation itself is the D operand: CK's D tensors are element tensors, the 1-bit pass masks cannot be one)
    plain     the same CK configurations without epilogue (how good is the CK kernel itself against MIOpen's pick)

and prints per layer the best configuration, the times, and the projected change of one attack iteration (every convolution
of the 12 non-projection blocks counted with its multiplicity).  Needs tools/bin/libck_probe.so (tools/ck_probe/build.sh).

    python tools/ck_conv_probe.py [--batch 125] [--json out.json]"""
import argparse
import ctypes
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transferattack_amd import _hip  # noqa: E402

CL = torch.channels_last


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / reps


class CK:
    def __init__(self):
        self.lib = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libck_probe.so"))
        self.lib.ckp_name.restype = ctypes.c_char_p
        vp, i = ctypes.c_void_p, ctypes.c_int
        self.lib.ckp_run.argtypes = [i, i, i, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp]

    def run(self, kind, one, idx, a, w, d0, d1, e, n, c, hi, wi, k, y, x, stride, pad):
        p = lambda t: None if t is None else t.data_ptr()      # noqa: E731
        return self.lib.ckp_run(kind, one, idx, p(a), p(w), p(d0), p(d1), p(e), n, c, hi, wi, k, y, x, stride, pad,
                                torch.cuda.current_stream().cuda_stream)

    def best(self, kind, one, *args):
        """-> (best us, name, index) over the configurations that take the problem"""
        out = (float("inf"), None, -1)
        for idx in range(self.lib.ckp_count(kind, one)):
            if self.run(kind, one, idx, *args) != 0:
                continue
            torch.cuda.synchronize()
            us = timeit(lambda: self.run(kind, one, idx, *args), reps=6, warm=1)
            if us < out[0]:
                out = (us, self.lib.ckp_name(kind, one, idx).decode(), idx)
        return out


def nhwc(*shape):
    return torch.randn(*shape, device="cuda").contiguous(memory_format=CL)


def probe_layer(ck, n, cin, cout, hw, ksize, epilogue):
    """one stride-1 convolution of a bottleneck; epilogue: 'bias' (conv1 / conv2) or 'bias_add' (conv3)"""
    pad = ksize // 2
    one = 1 if ksize == 1 else 0
    x, w = nhwc(n, cin, hw, hw), nhwc(cout, cin, ksize, ksize) * 0.05
    bias = torch.randn(cout, device="cuda")
    shortcut = nhwc(n, cout, hw, hw) if epilogue == "bias_add" else None
    geom = (n, cin, hw, hw, cout, ksize, ksize, 1, pad)
    rec = {"conv": "%dx%d %d->%d @%d" % (ksize, ksize, cin, cout, hw), "epilogue": epilogue,
           "GFLOP": 2.0 * n * hw * hw * cin * cout * ksize * ksize / 1e9}
    # ---- forward, unfused: MIOpen + glue (with the pass bits the backward of the product reads)
    y = F.conv2d(x, w, None, 1, pad)
    mask = _hip.pass_bits_like(y)
    t_conv = timeit(lambda: F.conv2d(x, w, None, 1, pad))
    if epilogue == "bias":
        t_glue = timeit(lambda: _hip.bias_act_(y, bias, mask=mask))
        ref = torch.clamp_min(F.conv2d(x, w, None, 1, pad) + bias.view(1, -1, 1, 1), 0)
    else:
        t_glue = timeit(lambda: _hip.bias_add_relu_(y, bias, shortcut, mask=mask))
        ref = torch.clamp_min((F.conv2d(x, w, None, 1, pad) + bias.view(1, -1, 1, 1)) + shortcut, 0)
    out = torch.empty_like(ref)
    kind = 1 if epilogue == "bias" else 2
    t_fused, name, idx = ck.best(kind, one, x, w, bias, shortcut, out, *geom)
    if idx >= 0:
        out.fill_(float("nan"))
        ck.run(kind, one, idx, x, w, bias, shortcut, out, *geom)
        rec["fwd_max_err"] = float((out - ref).abs().max() / ref.abs().max())
    t_plain, plain_name, _ = ck.best(0, one, x, w, None, None, out, *geom)
    rec["fwd"] = {"miopen_conv_us": round(t_conv, 1), "glue_us": round(t_glue, 1), "unfused_us": round(t_conv + t_glue, 1),
                  "ck_fused_us": round(t_fused, 1), "ck_fused": name, "ck_plain_us": round(t_plain, 1), "ck_plain": plain_name,
                  "miopen_TFLOPs": round(rec["GFLOP"] / t_conv / 1e3, 1), "ck_fused_TFLOPs": round(rec["GFLOP"] / t_fused / 1e3, 1)}
    # ---- backward data.  conv1's input gradient meets the junction (add + threshold on the block input); conv2 / conv3's input
    # gradient gets the threshold of the activation in front of it
    g = nhwc(n, cout, hw, hw)
    act = torch.relu(nhwc(n, cin, hw, hw))
    bits = _hip.pass_bits_like(act)
    _hip.bias_act_(act.clone(memory_format=torch.preserve_format), torch.zeros(cin, device="cuda"), mask=bits)
    spec = ([1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])
    bwd = lambda: torch.ops.aten.convolution_backward(g, x, w, None, *spec)[0]      # noqa: E731
    dx = bwd()
    t_bconv = timeit(bwd)
    junction = epilogue == "bias" and ksize == 1            # conv1 of a block
    other = nhwc(n, cin, hw, hw) if junction else None
    t_bglue = timeit(lambda: _hip.relu_mask(dx, act, dx, gb=other, mask=bits))
    ref = torch.ops.aten.threshold_backward(bwd() + other if junction else bwd(), act, 0)
    out = torch.empty_like(ref)
    kind = 5 if junction else 4
    d0, d1 = (other, act) if junction else (act, None)
    t_bfused, bname, bidx = ck.best(kind, one, g, w, d0, d1, out, *geom)
    if bidx >= 0:
        out.fill_(float("nan"))
        ck.run(kind, one, bidx, g, w, d0, d1, out, *geom)
        rec["bwd_max_err"] = float((out - ref).abs().max() / ref.abs().max())
    t_bplain, bplain_name, _ = ck.best(3, one, g, w, None, None, out, *geom)
    rec["bwd"] = {"miopen_conv_us": round(t_bconv, 1), "glue_us": round(t_bglue, 1), "unfused_us": round(t_bconv + t_bglue, 1),
                  "ck_fused_us": round(t_bfused, 1), "ck_fused": bname, "ck_plain_us": round(t_bplain, 1), "ck_plain": bplain_name,
                  "epilogue": "add + threshold (junction)" if junction else "threshold"}
    return rec


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=125)
    p.add_argument("--json", default="")
    args = p.parse_args()
    torch.backends.cudnn.benchmark = True
    _hip.load()
    ck = CK()
    n = args.batch
    # (planes, spatial size, stride-1 blocks whose three convolutions have these shapes): ResNet-50's layer1..4
    stages = [(64, 56, 3), (128, 28, 3), (256, 14, 5), (512, 7, 2)]
    rows, saved_us = [], 0.0
    for planes, hw, blocks in stages:
        for cin, cout, ks, epi in ((4 * planes, planes, 1, "bias"), (planes, planes, 3, "bias"), (planes, 4 * planes, 1, "bias_add")):
            rec = probe_layer(ck, n, cin, cout, hw, ks, epi)
            rec["blocks"] = blocks
            delta = (rec["fwd"]["unfused_us"] - rec["fwd"]["ck_fused_us"]) + (rec["bwd"]["unfused_us"] - rec["bwd"]["ck_fused_us"])
            rec["saved_us_per_iteration"] = round(blocks * delta, 1)
            saved_us += blocks * delta
            rows.append(rec)
            print(json.dumps(rec), flush=True)
    summary = {"batch": n, "projected_saving_ms_per_iteration": round(saved_us / 1e3, 3),
               "note": "sum over the stride-1 bottleneck convolutions of (MIOpen + glue) - (best CK configuration with the epilogue), "
                       "forward and backward data, each counted with its multiplicity; negative = the CK form is slower"}
    print(json.dumps(summary), flush=True)
    if args.json:
        json.dump({"layers": rows, "summary": summary}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
