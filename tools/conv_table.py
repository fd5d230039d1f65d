#!/usr/bin/env python
"""FLOPs and TFLOP/s of every convolution launch of one steady-state iteration (no GPU needed): aligns the conv-class kernels of
``one_iteration`` in a tools/steady_trace.py JSON with the convolutions the fused ResNet-50 path issues -- forward in module order
(per block conv1, conv2, conv3, projection), backward block by block from the last (conv3, conv2, conv1, projection), the stem's
backward on csrc/stem.hip -- and prints them sorted by time, plus the per-shape totals.

    python tools/conv_table.py profiles/r03/steady_state_b125_fused_glue_stem_r3j.json [batch]
"""
import json
import sys


def resnet50_convs():
    """[(name, cin, cout, k, stride, hin)] in forward order"""
    convs = [("stem 7x7/2", 3, 64, 7, 2, 224)]
    cin, h = 64, 56
    for li, (width, depth) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for j in range(depth):
            stride = 2 if (j == 0 and li > 0) else 1
            tag = "layer%d.%d" % (li + 1, j)
            convs.append((tag + ".conv1 1x1", cin, width, 1, 1, h))
            convs.append((tag + ".conv2 3x3" + ("/2" if stride == 2 else ""), width, width, 3, stride, h))
            hout = h // stride
            convs.append((tag + ".conv3 1x1", width, 4 * width, 1, 1, hout))
            if j == 0:
                convs.append((tag + ".proj 1x1" + ("/2" if stride == 2 else ""), cin, 4 * width, 1, stride, h))
            cin, h = 4 * width, hout
    return convs


def flops(n, cin, cout, k, stride, hin):
    hout = hin // stride
    return 2.0 * n * cout * hout * hout * cin * k * k


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 125
    seq = json.load(open(path))["one_iteration"]
    convs = resnet50_convs()
    launches = [k for k in seq if any(t in k["name"].lower() for t in ("igemm", "conv", "winograd"))
                and "ta::" not in k["name"] and "SubTensor" not in k["name"]]
    fwd = convs
    blocks, i = [], 1
    while i < len(convs):                                   # group per block to build the backward order
        size = 4 if (i + 3 < len(convs) and "proj" in convs[i + 3][0]) else 3
        blocks.append(convs[i:i + size])
        i += size
    bwd = []
    for blk in reversed(blocks):
        bwd += [blk[2], blk[1], blk[0]] + ([blk[3]] if len(blk) == 4 else [])
    expected = [("fwd",) + c for c in fwd] + [("bwd",) + c for c in bwd]
    stem_bwd = [k for k in seq if "stem7s2_input_grad" in k["name"]]
    if len(launches) != len(expected):
        print("warning: %d convolution launches in the trace, %d expected -- alignment by position may be off" % (len(launches), len(expected)))
    rows = []
    for k, e in zip(launches, expected):
        f = flops(n, *e[2:])
        rows.append((k["us"], e[0], e[1], f, f / k["us"] / 1e6, k["name"][:48]))
    if stem_bwd:
        f = flops(n, 3, 64, 7, 2, 224)
        rows.append((stem_bwd[0]["us"], "bwd", "stem 7x7/2 (csrc/stem.hip)", f, f / stem_bwd[0]["us"] / 1e6, "ta::stem7s2_input_grad_kernel"))
    total_us, total_f = sum(r[0] for r in rows), sum(r[3] for r in rows)
    print("convolutions of one iteration at batch %d: %d launches, %.2f ms, %.2f TFLOP -> %.1f TFLOP/s (fp32 matrix peak 157.3)"
          % (n, len(rows), total_us / 1e3, total_f / 1e12, total_f / total_us / 1e6))
    print("\n%-8s %-4s %-34s %9s %9s  %s" % ("us", "pass", "layer", "GFLOP", "TFLOP/s", "kernel"))
    for us, p, name, f, tf, kern in sorted(rows, reverse=True)[:24]:
        print("%8.1f %-4s %-34s %9.1f %9.1f  %s" % (us, p, name, f / 1e9, tf, kern))
    shapes = {}
    for us, p, name, f, tf, kern in rows:
        key = (p, name.split(" ", 1)[1] if " " in name else name, name.split(".")[0])
        agg = shapes.setdefault(key, [0.0, 0.0, 0])
        agg[0] += us
        agg[1] += f
        agg[2] += 1
    print("\nper (pass, kind, stage): launches, ms, TFLOP/s")
    for key, (us, f, cnt) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
        print("  %-4s %-12s %-8s %3d  %7.2f ms  %6.1f TFLOP/s" % (key[0], key[1][:12], key[2], cnt, us / 1e3, f / us / 1e6))


if __name__ == "__main__":
    main()
