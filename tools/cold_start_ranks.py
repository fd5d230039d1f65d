#!/usr/bin/env python
"""Cold start of ``bench.py --gpus N`` (review r5 item 3) on the ONE GPU a gpurun box has.

bench.py runs MIOpen in find mode; on a fresh node N ranks would all enter the find at once.  ``bench.staged_warmup`` lets
rank 0 warm up first.  This probe starts bench.py as TWO ranks that both drive device 0 (``--share-device 1 --backend gloo``:
the process group is only there for the ordering, the reporting collectives run on host tensors), each run with its OWN empty
MIOpen user database and kernel cache (MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR), i.e. a fresh node as far as MIOpen is
concerned, and prints one JSON line per run:

    one        one process, fresh databases                      cold(1)
    one:warm   one process on the databases ``one`` left          warm(1)
    staged     two ranks, rank 0 warms up first (the default)     expected ~ cold(1) + warm(1)
    together   two ranks, TA_BENCH_STAGED_WARMUP=0                 two contending cold starts
    staged8    EIGHT ranks on the one device, rank 0 first         the launch the driver's 8-GPU run makes, minus RCCL

    python tools/cold_start_ranks.py [--runs one,one:warm,staged,together] [--batch 125]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tag, ranks, dbs, extra_env, args):
    env = dict(os.environ, MIOPEN_USER_DB_PATH=os.path.join(dbs, "userdb"), MIOPEN_CUSTOM_CACHE_DIR=os.path.join(dbs, "cache"),
               TA_ALLOW_RANDOM_INIT="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.update(extra_env)
    os.makedirs(env["MIOPEN_USER_DB_PATH"], exist_ok=True)
    os.makedirs(env["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "1", "--batch", str(args.batch),
             "--cpu-images", "0", "--kernel-sweep", "0", "--literal-steps", "0"]
    if ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr",
               "127.0.0.1", "--master-port", str(args.port)] + bench + ["--share-device", "1", "--backend", "gloo"]
    else:
        cmd = [sys.executable] + bench
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.timeout)
    wall = time.perf_counter() - t0
    line = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
    cfg = line.get("config", {})
    return {"run": tag, "ranks": ranks, "process_wall_s": round(wall, 1), "rc": r.returncode, "startup": cfg.get("startup"),
            "images_per_s_per_rank": cfg.get("images_per_s_per_rank"), "value": line.get("value"),
            "stderr_tail": r.stderr[-400:] if r.returncode else ""}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--runs", default="one,one:warm,staged,together")
    p.add_argument("--batch", type=int, default=125)
    p.add_argument("--root", default="/tmp/ta_cold_ranks")
    p.add_argument("--port", type=int, default=29531)
    p.add_argument("--timeout", type=int, default=1500)
    args = p.parse_args()
    for spec in args.runs.split(","):
        name, _, warm = spec.partition(":")
        dbs = os.path.join(args.root, "db_" + name)
        if not warm:
            shutil.rmtree(dbs, ignore_errors=True)
        ranks = 1 if name == "one" else (8 if name == "staged8" else 2)
        extra = {"TA_BENCH_STAGED_WARMUP": "0"} if name == "together" else {}
        tag = {"one": "one process", "staged": "two ranks on one device, rank 0 warms up first",
               "staged8": "eight ranks on one device, rank 0 warms up first",
               "together": "two ranks on one device, both start cold together"}[name]
        tag += ", on the databases the previous run left" if warm else ", fresh MIOpen databases"
        print(json.dumps(run(tag, ranks, dbs, extra, args)), flush=True)


if __name__ == "__main__":
    main()
