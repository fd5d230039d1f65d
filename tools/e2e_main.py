#!/usr/bin/env python
"""End-to-end throughput of main.py on 1000 synthetic PNGs (/root/reference/main.py:43-53):
PNG decode -> upload -> attack (K = 10) -> quantise -> download -> PNG encode, everything the reference's ``main.py``
does per image, timed on the GPU box.  Writes the dataset under /tmp (labels.csv + images/*.png, utils.py:108-153), runs
main.py in this process with --profile for each arrangement and prints one JSON line per run.

    python tools/e2e_main.py [--images 1000] [--attack mifgsm] [--model resnet50]
"""
import argparse
import contextlib
import csv
import io
import json
import os
import shutil
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_dataset(root, count):
    from PIL import Image
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    g = torch.Generator().manual_seed(0)
    with open(os.path.join(root, "labels.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["filename", "label", "targeted_label"])
        for lo in range(0, count, 100):
            px = torch.randint(0, 256, (min(100, count - lo), 224, 224, 3), generator=g, dtype=torch.uint8).numpy()
            for j in range(px.shape[0]):
                Image.fromarray(px[j]).save(os.path.join(root, "images", "%04d.png" % (lo + j)))
                w.writerow(["%04d.png" % (lo + j), (lo + j) % 1000, (lo + j + 1) % 1000])


def run(data, out, attack, model, batchsize, extra, env):
    import importlib
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    shutil.rmtree(out, ignore_errors=True)
    import main as cli
    importlib.reload(cli)
    sys.argv = ["main.py", "--input_dir", data, "--output_dir", out, "--attack", attack, "--model", model, "--batchsize",
                str(batchsize), "--profile"] + extra
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        cli.main()
    wall = time.perf_counter() - t0
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    line = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
    line["process_wall_s_incl_model_build"] = round(wall, 2)
    line["written"] = len(os.listdir(out))
    return line


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--images", type=int, default=1000)
    p.add_argument("--attack", default="mifgsm")
    p.add_argument("--model", default="resnet50")
    p.add_argument("--root", default="/tmp/ta_e2e")
    args = p.parse_args()
    data = os.path.join(args.root, "data")
    if not os.path.isfile(os.path.join(data, "labels.csv")):
        t0 = time.perf_counter()
        write_dataset(data, args.images)
        print(json.dumps({"dataset": "%d synthetic 224x224 RGB PNGs (uniform noise: the worst case for the PNG codec)" % args.images,
                          "write_s": round(time.perf_counter() - t0, 1)}), flush=True)
    torch.backends.cudnn.benchmark = True
    fast = {"TA_FOLD_BN": "1", "TA_CHANNELS_LAST": "1", "TA_CK_EPILOGUE": "0"}
    plain = {"TA_FOLD_BN": "0", "TA_CHANNELS_LAST": "0", "TA_CK_EPILOGUE": "0"}
    runs = [("warm-up (MIOpen find)", 32, ["--coalesce", "4"], fast),
            ("reference batches of 32, one per device batch, reference-literal surrogate", 32, ["--coalesce", "1"], plain),
            ("reference batches of 32, one per device batch, folded BN + NHWC", 32, ["--coalesce", "1"], fast),
            ("reference batches of 32, four per device batch, folded BN + NHWC", 32, ["--coalesce", "4"], fast),
            ("same, 16 io threads", 32, ["--coalesce", "4", "--io_threads", "16"], fast),
            ("warm-up (composable_kernel site tuning: the decisions persist in ~/.cache/transferattack_amd)", 32,
             ["--coalesce", "4", "--io_threads", "16"], dict(fast, TA_CK_EPILOGUE="1")),
            ("same, glue passes as composable_kernel convolution epilogues (TA_CK_EPILOGUE=1)", 32,
             ["--coalesce", "4", "--io_threads", "16"], dict(fast, TA_CK_EPILOGUE="1"))]
    for tag, bs, extra, env in runs:
        r = run(data, os.path.join(args.root, "adv"), args.attack, args.model, bs, extra, env)
        r["run"] = tag
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
