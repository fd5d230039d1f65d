#!/usr/bin/env python
"""Per-kernel timing of every HIP entry point on MI355X: mean launch time (HIP events on the launch stream, operands
rotating over several buffer sets) and achieved GB/s at the ALGORITHMIC bytes of SURVEY.md 8(d).  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

E = 3 * 224 * 224
DEV = "cuda"


def timed(fn, reps=20):
    for i in range(4):
        fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(reps):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    _hip.load()
    out = {}
    sizes = [int(v) for v in os.environ.get("TA_BENCH_N", "32,160").split(",")]
    for n in sizes:
        sets = [[torch.randn(n, 3, 224, 224, device=DEV) * 1e-3 for _ in range(4)] for _ in range(3)]
        big = [torch.empty(5 * n, 3, 224, 224, device=DEV) for _ in range(2)] if n == 32 else None
        big15 = [torch.empty(15 * n, 3, 224, 224, device=DEV) for _ in range(2)] if n == 32 else None
        w = torch.rand(15, 15, device=DEV)
        w = (w / w.sum()).contiguous()
        perm = torch.cat([torch.randperm(n) for _ in range(3)]).to(DEV)
        u8 = torch.empty((n, 224, 224, 3), dtype=torch.uint8, device=DEV)
        mean = torch.tensor([0.485, 0.456, 0.406], device=DEV)
        std = torch.tensor([0.229, 0.224, 0.225], device=DEV)
        ws = torch.empty(4 * n * 49, device=DEV)

        def rec(name, us, bytes_per_elem, flop_per_elem=0):
            d = {"us": round(us, 2), "GBps": round(bytes_per_elem * E * n / us / 1e3, 1)}
            if flop_per_elem:
                d["TFLOPs"] = round(flop_per_elem * E * n / us / 1e6, 2)
            out["n%d_%s" % (n, name)] = d

        rec("tim_conv15", timed(lambda i: _hip.depthwise_conv2d_same(sets[i % 3][0], sets[i % 3][1], w)), 8, 450)
        rec("dim_fwd", timed(lambda i: _hip.dim_fwd(sets[i % 3][0], sets[i % 3][1], 246, 237, 3, 5)), 8)
        rec("dim_bwd", timed(lambda i: _hip.dim_bwd(sets[i % 3][0], sets[i % 3][1], 246, 237, 3, 5)), 8)
        rec("vmi_neighbor_philox", timed(lambda i: _hip.vmi_neighbor(sets[i % 3][0], sets[i % 3][1], sets[i % 3][2],
                                                                      0.09, seed=1, offset=i)), 12)
        rec("vmi_neighbor_normalized_philox", timed(lambda i: _hip.vmi_neighbor_normalized(
            sets[i % 3][0], sets[i % 3][1], sets[i % 3][2], mean, std, 0.09, seed=1, offset=i)), 12)
        rec("normalize_bwd_accumulate", timed(lambda i: _hip.normalize_bwd_accumulate(sets[i % 3][0], sets[i % 3][1], std, False)), 12)
        rec("grad_accumulate", timed(lambda i: _hip.grad_accumulate(sets[i % 3][0], sets[i % 3][1], False)), 12)
        rec("variance_finalize", timed(lambda i: _hip.variance_finalize(sets[i % 3][0], sets[i % 3][1],
                                                                        sets[i % 3][2], 20)), 12)
        rec("axpy", timed(lambda i: _hip.axpy(sets[i % 3][0], sets[i % 3][1], 0.006, sets[i % 3][2])), 12)
        rec("quantize_u8", timed(lambda i: _hip.quantize_u8_nhwc(sets[i % 3][0], sets[i % 3][1], u8)), 9)
        rec("normalize_fwd", timed(lambda i: _hip.normalize_fwd(sets[i % 3][0], sets[i % 3][1], mean, std)), 8)
        rec("normalize_bwd_partials", timed(lambda i: _hip.load().ta_normalize_bwd(
            sets[i % 3][0].data_ptr(), sets[i % 3][1].data_ptr(), std.data_ptr(), None, ws.data_ptr(), n, 3, 224 * 224,
            torch.cuda.current_stream().cuda_stream)), 8)
        rec("momentum_hook", timed(lambda i: _hip.momentum(sets[i % 3][0], sets[i % 3][1], sets[i % 3][2], 1.0)), 12)
        rec("update_delta_hook", timed(lambda i: _hip.update_delta_linf(sets[i % 3][0], sets[i % 3][1], sets[i % 3][2],
                                                                        0.006, 0.06, sets[i % 3][3])), 16)
        rec("mi_update_two_launch", timed(lambda i: _hip.mi_update(sets[i % 3][0], sets[i % 3][1], sets[i % 3][1],
                                                                   sets[i % 3][2], sets[i % 3][3], 1.0, 0.006, 0.06)), 24)
        # surrogate glue on a layer1-sized NHWC activation map of this batch ([n, 256, 56, 56])
        acts = [torch.randn(n, 256, 56, 56, device=DEV).contiguous(memory_format=torch.channels_last) for _ in range(6)]
        bias = torch.randn(256, device=DEV)
        ea = acts[0].numel() / (E * n)                   # elements relative to an image batch, for the GB/s column
        rec("glue_bias_relu_nhwc", timed(lambda i: _hip.bias_act_(acts[i % 3], bias)), 8 * ea)
        rec("glue_bias_add_relu_nhwc", timed(lambda i: _hip.bias_add_relu_(acts[i % 3], bias, acts[3 + i % 3])), 12 * ea)
        rec("glue_relu_mask_nhwc", timed(lambda i: _hip.relu_mask(acts[i % 3], acts[3 + i % 3], acts[i % 3])), 12 * ea)
        rec("glue_relu_mask_add_nhwc", timed(lambda i: _hip.relu_mask(acts[i % 3], acts[3 + i % 3], acts[i % 3],
                                                                      gb=acts[3 + (i + 1) % 3])), 16 * ea)
        # the same passes with pass bits: the forward writes 1 bit per element extra, the backward reads 1 bit instead of 4 B
        bits = _hip.pass_bits_like(acts[0])
        rec("glue_bias_relu_bits_nhwc", timed(lambda i: _hip.bias_act_(acts[i % 3], bias, mask=bits)), 8 * ea)
        rec("glue_relu_mask_bits_nhwc", timed(lambda i: _hip.relu_mask(acts[i % 3], acts[3 + i % 3], acts[i % 3], mask=bits)), 12 * ea)
        rec("glue_relu_mask_add_bits_nhwc", timed(lambda i: _hip.relu_mask(acts[i % 3], acts[3 + i % 3], acts[i % 3],
                                                                           gb=acts[3 + (i + 1) % 3], mask=bits)), 16 * ea)
        del acts, bits
        # the stem convolution's input gradient (7x7 / stride 2 / 3 -> 64): csrc/stem.hip against MIOpen's backward-data
        wst = (torch.randn(64, 3, 7, 7, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last)
        dys = [torch.randn(n, 64, 112, 112, device=DEV).contiguous(memory_format=torch.channels_last) for _ in range(2)]
        xs = torch.empty(n, 3, 224, 224, device=DEV)
        w2 = _hip.stem7s2_prepare(wst)
        flop_per_elem = 2 * 64 * 49 / 4                  # per element of dx: 49 taps / 4 phases x 64 channels, multiply + add
        rec("stem7s2_input_grad_mfma", timed(lambda i: _hip.stem7s2_input_grad(dys[i % 2], w2, xs), reps=10), 4 + 4 * 64 / 12, flop_per_elem)
        torch.backends.cudnn.benchmark = True
        rec("stem7s2_input_grad_miopen", timed(lambda i: torch.ops.aten.convolution_backward(
            dys[i % 2], xs, wst, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [True, False, False]), reps=10), 4 + 4 * 64 / 12,
            flop_per_elem)
        del dys, xs
        if n == 32:
            import random
            import numpy as np
            from transferattack_amd.transforms import bsr_draw
            random.seed(1); np.random.seed(1); torch.manual_seed(1)
            plan = torch.from_numpy(bsr_draw((n, 3, 224, 224), 3, 20)).to(DEV)
            stack = torch.empty(20 * n, 3, 224, 224, device=DEV)
            rec("bsr_fwd_x20", timed(lambda i: _hip.bsr_fwd(sets[i % 3][0], plan, stack, 20, 3), reps=6), 4 + 4 * 20)
            rec("bsr_bwd_x20", timed(lambda i: _hip.bsr_bwd(stack, plan, sets[i % 3][1], 20, 3), reps=6), 4 * 20 + 4)
            del stack
            from transferattack_amd.transforms import SIA_NOISE, sia_draw
            ns = 16
            splan, _ = sia_draw((ns, 3, 224, 224), 3, 20, None)
            splan = torch.from_numpy(splan).to(DEV)
            sx = sets[0][0][:ns].contiguous()
            sstack = torch.empty(20 * ns, 3, 224, 224, device=DEV)
            sg = torch.empty(ns, 3, 224, 224, device=DEV)
            es = ns / n                                        # relative to this batch, for the GB/s column
            rec("sia_fwd_x20_n16", timed(lambda i: _hip.sia_fwd(sx, splan, sstack, 20, 3, SIA_NOISE, seed=1, offset=i), reps=6),
                (4 + 4 * 20) * es)
            rec("sia_bwd_x20_n16", timed(lambda i: _hip.sia_bwd(sstack, splan, sx, sg, 20, 3, SIA_NOISE, seed=1, offset=i), reps=6),
                (4 * 20 + 8) * es)
            del sstack
            members = [sets[k][0] for k in range(3)] + [sets[0][1]]
            rec("sum_members_x4", timed(lambda i: _hip.sum_members(members, sets[i % 3][2])), 20)
        if big is not None:
            rec("sim_fwd_x5", timed(lambda i: _hip.scale_copies_fwd(sets[i % 3][0], big[i % 2], 5)), 24)
            rec("sim_bwd_x5", timed(lambda i: _hip.scale_copies_bwd(big[i % 2], sets[i % 3][0], 5)), 24)
            rec("admix_fwd_x15", timed(lambda i: _hip.admix_fwd(sets[i % 3][0], perm, big15[i % 2], 3, 5, 0.2)), 76)
            rec("admix_bwd_x15", timed(lambda i: _hip.admix_bwd(big15[i % 2], sets[i % 3][0], 3, 5)), 64)
        del sets, big, big15
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
