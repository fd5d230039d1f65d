#!/usr/bin/env python
"""Cold start of main.py (review item: a fresh box spent 250-285 s in MIOpen's find before 2.6 s of work).

Every run is a NEW process of ``main.py`` over the 1000 synthetic PNGs of tools/e2e_main.py with its own MIOpen user
database (MIOPEN_USER_DB_PATH: find-db / perf-db) and kernel cache (MIOPEN_CUSTOM_CACHE_DIR: compiled code objects), so
"fresh" means what a fresh box means -- no gfx950 find-db or kernel database ships with ROCm 7.2
(/opt/rocm/share/miopen/db holds none).  Modes:

    immediate   torch.backends.cudnn.benchmark = False: MIOpen's immediate mode (heuristic pick, only the chosen solver is built)
    find        benchmark = True: exhaustive find (every applicable solver built and timed) -- bench.py's setting
    fast        benchmark = True + MIOPEN_FIND_MODE=FAST: find-db hit or immediate-mode fallback, no search
    shipped     a process that starts from the databases another run left (what shipping them with the package buys)

    python tools/cold_start.py [--modes immediate,immediate:warm,fast,find,find:warm] [--keep DIR]
prints one JSON line per run: process wall seconds, main.py's own end-to-end images/s, database sizes."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tree_bytes(path):
    total = 0
    for base, _, files in os.walk(path):
        for f in files:
            try:
                total += os.path.getsize(os.path.join(base, f))
            except OSError:
                pass
    return total


def run(tag, data, out, dbs, env_extra, args):
    env = dict(os.environ, TA_FOLD_BN="1", TA_CHANNELS_LAST="1", TA_ALLOW_RANDOM_INIT="1", PYTHONPATH=ROOT,
               MIOPEN_USER_DB_PATH=os.path.join(dbs, "userdb"), MIOPEN_CUSTOM_CACHE_DIR=os.path.join(dbs, "cache"),
               TA_MIOPEN_DB="0")
    env.update(env_extra)
    os.makedirs(env["MIOPEN_USER_DB_PATH"], exist_ok=True)
    os.makedirs(env["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)
    shutil.rmtree(out, ignore_errors=True)
    cmd = [sys.executable, os.path.join(ROOT, "main.py"), "--input_dir", data, "--output_dir", out, "--attack", args.attack,
           "--model", args.model, "--batchsize", "32", "--profile", "--io_threads", "16"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.timeout)
    wall = time.perf_counter() - t0
    line = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
    return {"run": tag, "process_wall_s": round(wall, 1), "rc": r.returncode,
            "main_py_wall_s": line.get("wall_s"), "end_to_end_images_per_s": line.get("end_to_end_images_per_s"),
            "attack_s": (line.get("stage_busy_seconds") or {}).get("attack on the GPU (K iterations; HIP events)"),
            "written": len(os.listdir(out)) if os.path.isdir(out) else 0,
            "userdb_bytes": tree_bytes(env["MIOPEN_USER_DB_PATH"]), "kernel_cache_bytes": tree_bytes(env["MIOPEN_CUSTOM_CACHE_DIR"]),
            "env": {k: v for k, v in env_extra.items()}, "stderr_tail": r.stderr[-300:] if r.returncode else ""}


MODES = {"immediate": {"TA_CONV_TUNE": "0"},
         "find": {"TA_CONV_TUNE": "1"},
         "fast": {"TA_CONV_TUNE": "1", "MIOPEN_FIND_MODE": "FAST"},
         "hybrid": {"TA_CONV_TUNE": "1", "MIOPEN_FIND_MODE": "HYBRID"},
         "dynhybrid": {"TA_CONV_TUNE": "1", "MIOPEN_FIND_MODE": "DYNAMIC_HYBRID"}}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--modes", default="immediate,immediate:warm,fast,fast:warm")
    p.add_argument("--images", type=int, default=1000)
    p.add_argument("--attack", default="mifgsm")
    p.add_argument("--model", default="resnet50")
    p.add_argument("--root", default="/tmp/ta_cold")
    p.add_argument("--keep", default="", help="copy the databases of every mode here afterwards (e.g. gpurun_out/<tag>/miopen)")
    p.add_argument("--seed-from", default="", help="start every cold run from a copy of this database directory (userdb/ + cache/)")
    p.add_argument("--timeout", type=int, default=900)
    args = p.parse_args()
    data = os.path.join(args.root, "data")
    if not os.path.isfile(os.path.join(data, "labels.csv")):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from e2e_main import write_dataset
        write_dataset(data, args.images)
    for spec in args.modes.split(","):
        mode, _, warm = spec.partition(":")
        dbs = os.path.join(args.root, "db_" + mode)
        if not warm:
            shutil.rmtree(dbs, ignore_errors=True)
            if args.seed_from:
                shutil.copytree(args.seed_from, dbs)
        tag = "%s, %s" % (mode, "second process on the first one's databases" if warm else
                          ("databases seeded from %s" % args.seed_from if args.seed_from else "fresh databases"))
        print(json.dumps(run(tag, data, os.path.join(args.root, "adv"), dbs, MODES[mode], args)), flush=True)
        if args.keep and not warm:
            dst = os.path.join(args.keep, mode)
            shutil.rmtree(dst, ignore_errors=True)
            if tree_bytes(dbs) < 40 * 1024 * 1024:
                shutil.copytree(dbs, dst)
            else:
                os.makedirs(dst, exist_ok=True)
                shutil.copytree(os.path.join(dbs, "userdb"), os.path.join(dst, "userdb"))
                open(os.path.join(dst, "KERNEL_CACHE_TOO_LARGE.txt"), "w").write("%d bytes\n" % tree_bytes(os.path.join(dbs, "cache")))


if __name__ == "__main__":
    main()
